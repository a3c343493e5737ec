mkdir -p gpurun_out/v1
rocm-smi --showclocks --showpower 2>/dev/null | head -20 > gpurun_out/v1/smi.txt
timeout 300 tools/probe_mx/mx_probe > gpurun_out/v1/mx_probe.txt 2>&1
timeout 900 bash tools/probe_x8half/run.sh > gpurun_out/v1/x8half.txt 2>&1
tail -40 gpurun_out/v1/x8half.txt
