cd /root/repo
for t in 256 384 512 640; do echo "target $t"; python tools/probe_sweep_conv.py 8 256 5 rows_target_wgs=$t jacobi_cross_only=1 2>&1 | tail -2 | cut -c1-40; done
echo "512 no cross"; python tools/probe_sweep_conv.py 8 256 5 rows_target_wgs=512 2>&1 | tail -2 | cut -c1-40
echo "strip 512"; OPTS=rows_target_wgs=512 python tools/probe_strip.py 16384 32,64 2>&1 | grep rows768 | cut -c1-100
echo "U01 state bench 512"; python bench.py --no-full-rank --no-other-configs --no-energy --no-cpu-baseline --no-live-traffic --no-serial-pass --opt rows_target_wgs=512 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'])"
echo "U01 state bench 768"; python bench.py --no-full-rank --no-other-configs --no-energy --no-cpu-baseline --no-live-traffic --no-serial-pass 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'])"
