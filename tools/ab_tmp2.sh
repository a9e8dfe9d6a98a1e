OUT=gpurun_out/rm5.txt
: > $OUT
python - >> $OUT 2>&1 <<'PY'
from ipp_rl_amd import EngineConfig
from ipp_rl_amd.engine import IPPEngine
for dim, kw in ((50, dict(fixed_prior=True)), (50, {}), (100, dict(rank_cap=144)), (200, dict(rank_cap=144, node_capacity=64))):
    e = IPPEngine(EngineConfig(x_dim=dim, y_dim=dim), capacity=8, state="factor", window_rows=-1, **kw)
    print(dim, kw, "window", e.info.window_rows, "step_lds_bytes", e.info.step_lds_bytes)
    e.close()
PY
timeout 600 python -m pytest tests/test_hip_rect_meta.py tests/test_hip_tree.py tests/test_hip_configs.py -q 2>&1 | tr -cd "[:print:]\n" | tail -3 >> $OUT
tr -cd "[:print:]\n" < $OUT | grep -v amdgpu | tail -30
