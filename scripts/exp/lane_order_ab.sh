cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
cp sound-spaces_amd/csrc/libss_hip.so /tmp/lib_on.so
(cd sound-spaces_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSS_AB_NO_LOADER_LANE_ORDER ss_hip.hip -o /tmp/lib_off.so 2>&1 | grep -E "error")
for v in off on; do cp /tmp/lib_$v.so sound-spaces_amd/csrc/libss_hip.so; echo "== lane order $v"; python -m pytest tests/test_wav_loader.py -m gpu -q -k "orders_its_scatter" 2>&1 | tail -4; done
python -m pytest tests/test_wav_loader.py tests/test_context.py tests/test_c_abi.py -m gpu -q 2>&1 | tail -3
