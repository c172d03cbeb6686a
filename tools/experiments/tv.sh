#!/bin/bash
# GPU box: kernel times of library variants without verification (tools/time_variants.py): tv.sh "" a b c   ("" = the default build)
for v in "$@"; do echo -n "${v:-default}: "; H264BSD_VARIANT=$v timeout 300 python tools/time_variants.py 2>&1 | tail -1; done
