cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export GPU_MAX_HW_QUEUES=16
for g in 1 2 4 8 4; do
echo -n "groups $g: "
timeout 400 python bench.py --groups $g --no-cpu-baseline --no-staggered --no-desync --no-argb --no-end-to-end --ramp-seconds 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['device_ms_per_step']; print(round(d['value']/1e6,1), d['ms_per_step'], {k: round(v,1) for k,v in r.items() if isinstance(v,(int,float))})"
done
