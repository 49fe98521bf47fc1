#!/bin/bash
# round 5: host facts of the GPU box, then the full-width large-trace commits and the byte-equal full-size proofs
mkdir -p gpurun_out
{
echo "nproc: $(nproc)"; echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "memory.max: $(cat /sys/fs/cgroup/memory.max 2>/dev/null)"
free -g; grep -m1 "model name" /proc/cpuinfo; grep -c avx2 /proc/cpuinfo | head -1; grep -o -m1 'avx512[a-z]*' /proc/cpuinfo | sort -u | tr '\n' ' '
} > gpurun_out/r05_host.txt 2>&1
python3 -m pytest tests/test_gpu_merkle.py -k "full_width" -x -q -s -m gpu > gpurun_out/r05_large_commit.log 2>&1
echo "commit rc=$?" >> gpurun_out/r05_large_commit.log
python3 -m pytest tests/test_gpu_prove.py -k "full_size" -x -q -s -m gpu > gpurun_out/r05_full_proofs.log 2>&1
echo "prove rc=$?" >> gpurun_out/r05_full_proofs.log
tail -5 gpurun_out/r05_large_commit.log gpurun_out/r05_full_proofs.log; cat gpurun_out/r05_host.txt
