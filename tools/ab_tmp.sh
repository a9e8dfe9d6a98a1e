export PYTHONPATH=$PWD
timeout 900 python - <<'PY'
import torch, bench, json, time
dev = torch.device('cuda:0')
for drv, kw in (("device", dict(roots=256, sims=64)), ("device", {}), ("device", {})):
    t0 = time.time()
    r = bench.run_mcts_driver(torch, dev, driver=drv, **kw)
    print(drv, kw, {k: r[k] for k in ('value','seconds_per_search','device_tree_steps','launches','nodes','inferences','all_policies_valid')}, 'wall', round(time.time()-t0,1))
PY
