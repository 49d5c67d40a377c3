mkdir -p gpurun_out/guard
L=$PWD/tools/bin/libguard_malloc.so
export CTM_ABORT_BACKTRACE=0
{
echo "== control: no shim, no caching"; PYTORCH_NO_HIP_MEMORY_CACHING=1 timeout 300 python tools/guard/probe.py 2>&1 | tail -25
echo "== shim, never guarding"; LD_PRELOAD=$L GUARD_MIN=1000000000000 PYTORCH_NO_HIP_MEMORY_CACHING=1 timeout 300 python tools/guard/probe.py 2>&1 | tail -25
for mode in end start; do for al in 16 256 4096; do
echo "== shim mode=$mode align=$al"; LD_PRELOAD=$L GUARD_MODE=$mode GUARD_ALIGN=$al PYTORCH_NO_HIP_MEMORY_CACHING=1 timeout 300 python tools/guard/probe.py 2>&1 | tail -25
done; done
echo "== shim + arena guard align=16"; LD_PRELOAD=$L CTM_ARENA_GUARD=1 PYTORCH_NO_HIP_MEMORY_CACHING=1 timeout 300 python tools/guard/probe.py 2>&1 | tail -25
} > gpurun_out/guard/probe.log 2>&1
cat gpurun_out/guard/probe.log
