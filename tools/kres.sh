#!/bin/bash
# Register / spill summary of every kernel of one HIP source: bash tools/kres.sh candidate_reranking_cir_amd/csrc/gemm256.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$(dirname $0)/../include "${@:2}" -c $1 -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name:/ {name=$(NF-1)} / VGPRs:/ {v=$(NF-1)} / SGPRs:/ {sg=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /VGPRs Spill/ {sp=$(NF-1)} /LDS Size/ {print name, "vgpr", v, "sgpr", sg, "scratch", sc, "vspill", sp, "lds", $(NF-1)}' | c++filt
