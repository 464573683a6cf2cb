"""GPU: K-eval alone on a large resident batch of config-4 candidates (the genuinely HBM-streaming kernel): ms, candidates/s,
algorithmic GB/s (2*RF*P + 2*RF_cur*P + B bytes per candidate).  Test tooling."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic
kao.init(0)
dev = torch.device("cuda", 0)
for cfg, ntop in ((4, 1), (2, 1)):
    t = synthetic.make_config(cfg, n_topics=ntop)[0]
    n = 1 << 18 if cfg == 4 else 1 << 16
    per = t.n_partitions * t.rf
    g = torch.Generator(device=dev); g.manual_seed(1)
    cand = torch.randint(0, t.n_brokers, (n, per), dtype=torch.int32, device=dev, generator=g).to(torch.int16)
    obj = torch.empty(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    plan = kao.EvalPlan(t)
    for _ in range(2):
        plan.run(cand.data_ptr(), n, obj.data_ptr()); plan.sync()
    ms = []
    for _ in range(7):
        plan.run(cand.data_ptr(), n, obj.data_ptr()); ms.append(plan.sync())
    plan.close()
    m = sorted(ms)[len(ms) // 2]
    by = n * (2 * t.rf * t.n_partitions + 2 * t.rf_cur * t.n_partitions + t.n_brokers)
    print(f"cfg{cfg}: {m * 1e3:.1f} us for {n} candidates = {n / (m * 1e-3):.3e} cand/s = {by / (m * 1e-3) / 1e9:.0f} GB/s algorithmic ({by / (m * 1e-3) / 8e12 * 100:.1f} % of 8 TB/s)")
