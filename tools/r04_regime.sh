cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 600 python tools/step_variants.py --no-probe --steps 100 --repeat 2 --json gpurun_out/r04/step_variants4.json default edge_after_enrich=1 sweep_exclusive=0 sweep_exclusive=0,edge_after_enrich=1 2>&1 | grep variant | cut -c1-150
for o in "" "--opt edge_after_enrich=1" "--opt sweep_exclusive=0" "--opt sweep_exclusive=0 --opt edge_after_enrich=1"; do
python bench.py --force-gather-path --no-cpu-baseline --no-legs --steps 100 $o 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gather path', d['options'], d['ms_per_step'], d['roofline']['frac'], d['gather']['ms_per_step_without_pack_and_gather'])"
done
