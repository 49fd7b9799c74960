# Another build of libdpb.so with extra compile flags, for same-session A/Bs through DPB_LIB (tools/ab_env.sh):
#   bash tools/build_variant.sh NAME "-DDPB_OUT_STORE=0 ..."      -> diffusion_pullback_amd/csrc/build/NAME/libdpb.so   (all files compiled in parallel)
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../diffusion_pullback_amd/csrc" || exit 1
mkdir -p build/$NAME
CF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $FLAGS"
pids=""
for f in gemm gemm_dma gemm_ring64 gemm_p8 gemm_wres gemm_halo norm attn attn_fused elementwise orth; do
  /opt/rocm/bin/hipcc $CF -c $f.hip -o build/$NAME/$f.o & pids="$pids $!"
done
/opt/rocm/bin/hipcc $CF -x hip -c engine.cpp -o build/$NAME/engine.o & pids="$pids $!"
for p in $pids; do wait $p || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/$NAME/libdpb.so build/$NAME/*.o && echo "built build/$NAME/libdpb.so"
