#!/bin/bash
# Probe the GPU box for a Go toolchain. Output kept in profiles/r02_go_probe.txt
mkdir -p gpurun_out
{
echo "== which go / gccgo"; which go gccgo gofmt 2>&1
echo "== go version"; go version 2>&1
echo "== ls"; ls -d /usr/local/go /usr/lib/go* /usr/lib/golang /opt/go* /root/go /snap/go 2>&1
echo "== find go binaries"; find / -xdev \( -name go -o -name gccgo -o -name 'go1.*' \) -type f 2>/dev/null | head
echo "== nproc"; nproc; lscpu | head -20
echo "== rocm-smi"; rocm-smi --showproductname 2>&1 | head -20
echo "== devices"; python -c "import torch; print(torch.cuda.device_count(), torch.cuda.get_device_name(0))"
} > gpurun_out/r02_go_probe.txt 2>&1
cat gpurun_out/r02_go_probe.txt
# round 3: any box that DOES have Go (and a reference tree: REF=...) closes SURVEY 8(c) by itself -- the determinised-reference job
bash oracle/run_ref.sh 2>&1 | tee gpurun_out/go_parity.txt
