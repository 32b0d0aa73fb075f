#!/usr/bin/env bash
# Builds kobato-eyes_amd/libkeyes_hip.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
src="$here/csrc"
out="$here/libkeyes_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off
       -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function)
"$HIPCC" "${FLAGS[@]}" ${KE_EXTRA_FLAGS:-} -o "$out" \
    "$src/ke_api.hip" "$src/ke_hash.hip" "$src/ke_scan.hip" "$src/ke_ssim.hip" "$src/ke_synth.hip" "$src/ke_comm.hip" "$src/ke_jpeg.hip" "$src/ke_png.hip" "$src/ke_bmp.hip" "$src/ke_gif.hip" "$src/ke_tiff.hip" "$src/ke_normalise.hip" \
    "$src/ke_coeffs.cpp" "$src/ke_cluster.cpp" -ldl
echo "built $out"
