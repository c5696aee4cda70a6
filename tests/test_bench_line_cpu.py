"""bench.py's printed line must stay small and flat (round-4 verdict: the driver's record keeps top-level scalars and the scalar
members of `config` / `roofline` / `cpu_baseline` plus the last ~2 KB of stdout — the nested ResNet-50 / TP-block objects of round 4
never reached BENCH_r04.json). flat_summary() is run here on round 4's own nested line (profiles/r04_bench_line.json)."""
import json
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def test_flat_summary_carries_the_metric_and_stays_small():
    import bench

    old = json.loads((REPO / "profiles" / "r04_bench_line.json").read_text())
    detail = {k: old.pop(k) for k in ("tp_block", "graph_resnet50", "extras") if k in old}
    assert detail.keys() == {"tp_block", "graph_resnet50", "extras"}
    bench.flat_summary(old, detail, 1)
    cfg, roof = old["config"], old["roofline"]
    assert cfg["resnet50_bs128_fp16_graph_ms"] == detail["graph_resnet50"]["hipgraph_ms"]
    assert cfg["bert_base_bs32_seq512_fp16_graph_ms"] == detail["graph_resnet50"]["bert_base_bs32_seq512_f16"]["hipgraph_ms"]
    assert cfg["llama7b_block_tp1_ms"] == detail["tp_block"]["ms_per_block"]
    assert roof["softmax_196608x512_f16_frac_hbm"] == detail["extras"]["softmax_196608x512_f16"]["frac_hbm_peak"]
    assert roof["layernorm_16384x768_f16_frac_hbm"] == detail["extras"]["layernorm_16384x768_f16"]["frac_hbm_peak"]
    assert roof["membound_hbm_min_row"] in detail["extras"]["membound"]["rows"] and 0 < roof["membound_hbm_min_frac"] < 1
    # every member of the kept objects the summary added is a scalar, and the whole line is < 3 KB without the nested
    # per-launch / cold20 objects' growth (they were there in round 4 already)
    for obj in (cfg, roof):
        for k, v in obj.items():
            if k in ("cold20", "kernel_us_per_launch"):
                continue
            assert not isinstance(v, (dict, list)), k
    assert len(json.dumps(old)) < 4096, len(json.dumps(old))
