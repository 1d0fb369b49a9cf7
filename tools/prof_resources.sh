#!/bin/bash
# Compiler resource report of every kernel (registers, spills, scratch, LDS,
# occupancy) -> stdout; no GPU needed.  profiles/rNN_kernel_resources.txt is
# this script's output for the round's final build.
cd "$(dirname "$0")/../libdeflate_amd/csrc"
for f in deflate_kernel deflate_small inflate_kernel inflate_stream checksum_kernels compact_kernels; do
  # (the flags per object: see the Makefile)
  x="-mllvm -disable-machine-licm"; case $f in deflate_kernel) x="$x -mllvm -amdgpu-sched-strategy=max-ilp";; checksum_kernels|compact_kernels) x=;; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden \
      -ffp-contract=off $x -Rpass-analysis=kernel-resource-usage -c $f.hip -o /dev/null 2>&1 |
    grep -E "Function Name|VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|SGPRs:" |
    sed 's/.*remark: *//; s/ \[-Rpass-analysis=kernel-resource-usage\]//' |
    awk '/Function Name/{if (l) print l; l=$0; next} {l=l " | " $0} END{print l}' | sed 's/  */ /g'
done
