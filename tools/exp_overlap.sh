mkdir -p gpurun_out/exp
B="python bench.py --signed --steps 1 --warmup 2 --no-cpu-baseline --no-serial-pass"
run() { tag=$1; shift; "$@" > gpurun_out/exp/$tag.json 2> gpurun_out/exp/$tag.err; python - <<PY
import json
d=json.loads(open('gpurun_out/exp/$tag.json').read().strip().splitlines()[-1])
print('$tag', round(d['ms_per_step'],1), d.get('svd'), (d.get('energy_parity') or {}))
PY
}
run base $B
run absT $B --opt lz_abs_accuracy=1
