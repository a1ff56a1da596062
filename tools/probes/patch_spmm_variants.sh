# A/B of patch-kernel variants on one box:  bash tools/probes/patch_spmm_variants.sh "-DSOME_SWITCH" "-DNONE"   -- each argument is a set of extra
# compiler flags: the patch kernel object is rebuilt with them, the library relinked, then the timing and the traffic probe run (boxes differ by
# +-5 %: only runs of ONE call compare).  The switches themselves live in the experiment, not in the tree (HISTORY section 10 lists what was tried).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R/stc-gnn_amd/csrc
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include $v -c stc_spmm_patch.hip -o stc_spmm_patch.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o $R/stc-gnn_amd/libstc_hip.so || exit 1
  echo "=== variant [$v] TILES=${TILES:-}"
  (cd $R && python tools/probes/patch_spmm_unit.py | grep -E "fwd patch|copy" && bash tools/probes/patch_spmm_traffic.sh | grep -E "spmm_patch")
done
