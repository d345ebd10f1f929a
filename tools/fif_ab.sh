set -o pipefail
mkdir -p gpurun_out/fif
for spp in 1024 128; do for f in 1 2 3; do
if [ $f = 3 ]; then export HT_BENCH_PRIO=1; ff=2; else unset HT_BENCH_PRIO; ff=$f; fi
timeout -k 10 200 python bench.py --no-cpu-baseline --spp $spp --frames-in-flight $ff --steps 20 --warmup 3 > gpurun_out/fif/d_${spp}_$f.json 2> gpurun_out/fif/d_${spp}_$f.err || exit 1
python -c "
import json;d=json.loads(open('gpurun_out/fif/d_${spp}_$f.json').read().strip().splitlines()[-1]);print('spp',$spp,'in flight',$ff,'prio' if $f==3 else '',d['value'],d['ms_per_step'])"
done; done
