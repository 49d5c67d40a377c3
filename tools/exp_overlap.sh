mkdir -p gpurun_out/exp
B="python bench.py --signed --steps 1 --warmup 2 --no-cpu-baseline --no-serial-pass"
run() { tag=$1; shift; "$@" > gpurun_out/exp/$tag.json 2> gpurun_out/exp/$tag.err; python - <<PY
import json
d=json.loads(open('gpurun_out/exp/$tag.json').read().strip().splitlines()[-1])
print('$tag', round(d['ms_per_step'],1))
PY
}
run queue $B
CTM_UNIT_STAGGER_MS=40 run st40 $B
CTM_UNIT_STAGGER_MS=40 run st40w256 $B --opt rows_target_wgs=256
CTM_UNIT_STAGGER_MS=40 run st40w512 $B --opt rows_target_wgs=512
CTM_LARGE_N_UNITS=4 CTM_UNIT_STAGGER_MS=25 run u4st25w256 $B --opt rows_target_wgs=256
CTM_LARGE_N_UNITS=4 run u4queue $B
CTM_LARGE_N_UNITS=3 run u3queue $B
