#!/bin/bash
# GPU box: repeat the 8-thread C caller with frame dumps until a decoder prints a wrong digest, then say where it differs
set -u
out=gpurun_out/threads; rm -rf $out; mkdir -p $out /tmp/dh
gcc -O2 -std=gnu11 -Iinclude tests/c_caller/decode_hash.c -o $out/dh -Lh264bsd_amd/lib -lh264bsd_mi355x -lpthread -Wl,-rpath,$PWD/h264bsd_amd/lib || exit 1
want=$(python -c "import json;print(json.load(open('tests/golden/golden.json'))['test_640x360']['sha256_all'])")
for i in $(seq 1 ${N:-150}); do
  DH_DUMP=/tmp/dh $out/dh -t 8 tests/golden/test_640x360.h264 2>$out/err.log > $out/o.txt
  if grep '^decoder' $out/o.txt | grep -vq "$want"; then
    echo "run $i: wrong digest"; grep '^decoder' $out/o.txt | grep -v "$want"
    python - <<'P'
import numpy as np, glob, json, hashlib
g = json.load(open('tests/golden/golden.json'))['test_640x360']
wmb, hmb = g['width_mbs'], g['height_mbs']; W, H = 16*wmb, 16*hmb; fb = wmb*hmb*384
fr = {f: np.fromfile(f, dtype=np.uint8).reshape(-1, fb) for f in sorted(glob.glob('/tmp/dh/dec*_pass0.yuv'))}
goodf = None
for f, a in fr.items():
    if all(hashlib.sha256(a[i].tobytes()).hexdigest() == g['frame_sha256'][i] for i in range(len(a))): goodf = a; break
for f, a in fr.items():
    wrong = [i for i in range(len(a)) if hashlib.sha256(a[i].tobytes()).hexdigest() != g['frame_sha256'][i]]
    if not wrong: continue
    print(f, 'wrong pictures (output order):', wrong)
    for i in wrong[:1]:
        d = a[i] != goodf[i]
        dy = d[:W*H].reshape(hmb,16,wmb,16)
        mbs = np.argwhere(dy.any(axis=(1,3)))
        print('  picture', i, 'bytes differing', int(d.sum()), 'luma MBs', len(mbs))
        for y, x in mbs[:60]:
            m = dy[y,:,x,:]
            rr = np.nonzero(m.any(axis=1))[0]; cc = np.nonzero(m.any(axis=0))[0]
            A = a[i][:W*H].reshape(H,W)[16*y:16*y+16,16*x:16*x+16].astype(int); B = goodf[i][:W*H].reshape(H,W)[16*y:16*y+16,16*x:16*x+16].astype(int)
            print(f'   MB ({y},{x}): rows {rr.min()}..{rr.max()} cols {cc.min()}..{cc.max()} n={int(m.sum())} maxdiff={int(np.abs(A-B).max())}')
P
    break
  fi
done
echo "runs: $i"
