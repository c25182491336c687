#!/bin/bash
# Where a step of the patch weight-gradient kernel goes: builds of conv_wgrad.hip with -DTEDSPAD_W3P_ABLATE=N (1 = no MFMAs, 2 = no fragment reads, 4 = no DMA; sums are
# wrong by construction) timed on the probe. Build here (CPU): bash scripts/w3p_ablate.sh build; run on the GPU box: bash scripts/w3p_ablate.sh
cd "${GRAFT_REPO_ROOT:-.}"
if [ "$1" = build ]; then
  mkdir -p ab
  for n in 1 2 3 4 6 7; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -I include -I ted_spad_amd/csrc -DTEDSPAD_W3P_ABLATE=$n -c ted_spad_amd/csrc/conv_wgrad.hip -o ab/w3p_abl$n.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/w3p_abl$n.so ab/w3p_abl$n.o $(ls ted_spad_amd/csrc/build/*.o | grep -v conv_wgrad.o)
  done
  exit 0
fi
cp ted_spad_amd/libtedspad_hip.so ab/keep.so
trap 'cp ab/keep.so ted_spad_amd/libtedspad_hip.so' EXIT
echo "== full"; timeout -k 10 200 python scripts/wgrad_probe.py 2>&1 | tail -9 | head -4
for n in 1 2 3 4 6 7; do
  cp ab/w3p_abl$n.so ted_spad_amd/libtedspad_hip.so
  echo "== ablate $n"; timeout -k 10 200 python scripts/wgrad_probe.py 2>&1 | tail -9 | head -4
done
