"""bench.py's host-side planning, without a GPU: the probe table the library will lay out (the load-factor chain of
ku_ctx_set_taxonomy) and where a routed step's counter traffic comes from (profiles/route_traffic.json) -- the
one-GPU configuration lines as profiled, and the 8-GPU line of the plain `--gpus 8` flow, which no box of the pool can run."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("ku_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_planned_table_follows_the_load_factor_chain():
    b = load_bench()
    free = 280e9
    tb, lf = b.planned_table_bytes(610_000_000, free)  # configs[1]: 80 B per pair fit 40 % of the free memory
    assert lf == 0.2 and abs(tb - 610_000_000 / 1.6 * 128) < 1e6
    tb, lf = b.planned_table_bytes(3_720_000_000, free - 45e9)  # one shard of configs[2]: 0.2 does not, 0.3 does
    assert lf == 0.3 and tb < 0.85 * (free - 45e9)
    tb, lf = b.planned_table_bytes(6_000_000_000, 200e9)  # 72 GB of pairs beside 200 GB of free memory: a dense table
    assert lf == 0.6 and tb == int(6_000_000_000 / 4.8) * 128
    tb, lf = b.planned_table_bytes(24_000_000_000, 10e9)  # nothing fits: the sorted layout
    assert lf is None and tb == 24_000_000_000 * 12


def test_routed_traffic_of_the_profiled_and_of_the_full_world(tmp_path):
    b = load_bench()
    path = os.path.join(ROOT, "profiles", "route_traffic.json")
    tj = json.load(open(path))
    rev = tj["kernel_rev"]
    ent = tj["workloads"]["nt15_species96000_shards8_ws1_reads10000000_len150"]["hbm_bytes_per_step"]
    # as profiled: one rank of eight shards doing everybody's scan and resolve -- the line models the rank's share (1 / 8)
    t, note = b.routed_traffic(path, rev, 15, 96000, 8, 1, 10_000_000, 150, 1 / 8)
    assert abs(t - ((ent["scan"] + ent["resolve"]) / 8 + ent["owner"])) < 1 and "r06_config2" in note
    # the world of eight holds the layout: the same bytes, taken from the one-rank profile
    t8, note8 = b.routed_traffic(path, rev, 15, 96000, 8, 8, 10_000_000, 150, 1.0)
    assert abs(t8 - t) < 1 and "1 / 8" in note8
    # other worlds, other workloads, other kernel sources: nothing is made up
    assert b.routed_traffic(path, rev, 15, 48000, 4, 4, 10_000_000, 150, 1.0)[0] is None
    assert b.routed_traffic(path, rev, 15, 96000, 8, 2, 10_000_000, 150, 0.25)[0] is None
    t, note = b.routed_traffic(path, "000000000000", 15, 96000, 8, 8, 10_000_000, 150, 1.0)
    assert t is None and "refused" in note
    assert b.routed_traffic(str(tmp_path / "missing.json"), rev, 15, 96000, 8, 8, 10_000_000, 150, 1.0)[0] is None
