cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
cp sound-spaces_amd/csrc/libss_hip.so /tmp/lib_2wg.so
(cd sound-spaces_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSS_FEAT_PROBE_1WG ss_hip.hip -o /tmp/lib_1wg.so 2>&1 | grep -E "error")
for rep in 1 2; do for v in 2wg 1wg; do cp /tmp/lib_$v.so sound-spaces_amd/csrc/libss_hip.so; echo "== $v"; python scripts/kbench_features.py --units 256 2>/dev/null; done; done
