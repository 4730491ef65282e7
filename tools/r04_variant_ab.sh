# A/B of whole-library builds under the default staged step: stock, every variants/lib_*.so, stock again (same box)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp emplanner_carla_amd/libemplanner.so /tmp/stock.so
run() { timeout 300 python tools/step_variants.py --no-probe --steps 100 --repeat 2 default sweep_exclusive=0 2>&1 | grep variant | cut -c1-125; python tools/dp_microbench.py 4096 2>&1 | grep "mode 1" | cut -c1-120; }
echo "== stock"; run
for v in variants/lib_*.so; do echo "== $v"; cp $v emplanner_carla_amd/libemplanner.so; run; done
cp /tmp/stock.so emplanner_carla_amd/libemplanner.so
echo "== stock again"; run
