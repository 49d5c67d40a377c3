cd /root/repo
for o in "" "--opt timing_min_flops=5e9" "--opt timing_min_flops=5e9 --opt lz_quad_exit=1e-7"; do
python bench.py --signed --no-other-configs --no-energy --no-cpu-baseline --no-live-traffic --no-serial-pass --steps 4 $o 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o', l['ms_per_step'], l['svd'])"
done
