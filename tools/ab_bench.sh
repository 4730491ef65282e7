# usage: ab.sh libA libB [bench args]: alternates two builds of the library through bench.py, three runs each
A=$1; B=$2; shift 2
for r in 1 2 3; do for L in $A $B; do cp $L emplanner_carla_amd/libemplanner.so; python bench.py --no-legs --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d.get('kernels_ms',{});print('$L',d['ms_per_step'],d['ms_per_step_min_max'],{n:round(v*1e3,1) for n,v in k.items()})"; done; done
