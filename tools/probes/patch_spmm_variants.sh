# build a variant of the patch kernel object with extra flags on the box, relink, run the traffic + timing probe
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R/stc-gnn_amd/csrc
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include $v -c stc_spmm_patch.hip -o stc_spmm_patch.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o $R/stc-gnn_amd/libstc_hip.so || exit 1
  echo "=== variant [$v] TILES=${TILES:-} ONE_ITEM=${STC_PATCH_ONE_ITEM:-}"
  (cd $R && python tools/probes/patch_spmm_unit.py | grep -E "fwd patch|copy" && bash tools/probes/patch_spmm_traffic.sh | grep -E "spmm_patch")
done
