"""Identity of the kernel sources for measurements that cannot be repeated inside bench.py (the PMC traffic table,
profiles/rNN_traffic.json): sha256 over kernels.hip.h, kernels/*.hip.h + framejob.h with comments and white space removed, so that a
table stays valid across comment edits and goes stale with any change of code."""
import hashlib
import os
import re

_FILES = ("kernels.hip.h", "kernels/common.hip.h", "kernels/k_dbk.hip.h", "kernels/k_copy.hip.h", "kernels/k_recon_inter.hip.h", "kernels/convert.hip.h",
          "kernels/tail_common.hip.h", "kernels/k_frame_intra.hip.h", "kernels/k_frame_dbk.hip.h", "kernels/k_pixels_io.hip.h", "framejob.h")


def _code_only(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return re.sub(r"\s+", " ", text).strip()


def kernel_source_sha256(root):
    h = hashlib.sha256()
    for f in _FILES:
        h.update(_code_only(open(os.path.join(root, "h264bsd_amd", "csrc", f)).read()).encode())
    return h.hexdigest()
