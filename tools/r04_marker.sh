cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 600 python tools/step_variants.py --no-probe --steps 100 --repeat 3 --json gpurun_out/r04/step_variants5.json default sweep_marker=0 2>&1 | grep variant | cut -c1-150
for o in "" "--opt sweep_marker=0"; do for r in 1 2; do
python bench.py --force-gather-path --no-cpu-baseline --no-legs --steps 100 $o 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gather path', d['options'], d['ms_per_step'], d['roofline']['frac'])"
done; done
bash tools/r04_trace.sh current | tail -28
