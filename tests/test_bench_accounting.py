"""CPU suite: bench.py's roofline accounting reproduces SURVEY.md 8(d)'s per-unit figures."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_algorithmic_bytes_match_survey_figures():
    import bench
    # B_alg = sum_{l>=1} 2s(N_{l-1}+N_l) + sum_l s(4N_l + 2N_{l+1}),  s = 8
    b = bench.algorithmic_bytes_per_pair(1920, 1080, 3)
    assert b == 149_299_200 and abs(b / (1920 * 1080) - 72.0) < 1e-9          # 72.0 B/px, 149.3 MB per 1080p pair
    assert abs(bench.algorithmic_bytes_per_pair(3840, 2160, 5) / (3840 * 2160) - 74.5) < 0.05   # 617.9 MB
    assert abs(bench.algorithmic_bytes_per_pair(1080, 1080, 4) / (1080 * 1080) - 74.0) < 0.05
    # dominant kernel (LK at level 0): read 2 images + coarse u,v, write u,v = 36 B/px
    assert bench.lk0_bytes_per_pair(1920, 1080, 3) == 36 * 1920 * 1080
    assert set(bench.CONFIGS) >= {"1080p", "4k", "8k", "pacman", "webcam"}
    # stream configs build P+1 pyramids for P pairs (not 2P): the pyramid term shrinks by (P+1)/(2P)
    full, stream = bench.algorithmic_bytes_per_pair(240, 160, 3), bench.algorithmic_bytes_per_pair(240, 160, 3, stream_pairs=244)
    pyr = 2 * 8 * ((240 * 160 + 120 * 80) + (120 * 80 + 60 * 40))
    assert abs((full - stream) - pyr * (1 - 245 / 488)) <= 1


def test_usable_cores_is_what_the_scheduler_allows():
    import bench
    cores, affinity, quota = bench.usable_cores()
    assert 1 <= cores <= affinity == len(os.sched_getaffinity(0))
    assert quota is None or quota > 0


def test_committed_pmc_traffic_is_usable_by_bench():
    """roofline.traffic / rocprof_avg_launch_ms come from profiles/pmc_traffic.json: one record per workload, each tagged with
    the digest of the kernel sources it was measured on.  On matching code (tree AND loaded library) the headline record must
    give a per-launch figure between the algorithmic bytes and twice that (aprons, halos); on any other code -- or workload --
    bench.py must report null rather than a stale number."""
    import json
    import bench
    cols, rows, levels, win, pairs = bench.CONFIGS["1080p"]
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
        top = json.load(fh)
    assert "workloads" in top
    rec = top["workloads"].get(f"{cols}x{rows}x{pairs}")
    assert rec, "no headline workload in profiles/pmc_traffic.json"
    names = [k for k in rec if k.startswith("lk::lk_") and ", true" in k]     # HAS_UP instantiations
    assert names, "no level-0 LK kernel in the headline record"
    ts = [bench.measured_traffic(k, cols, rows, pairs) for k in names]
    if rec.get("_csrc_sha") == bench.csrc_digest() == bench.loaded_lib_digest():
        sys.path.insert(0, os.path.join(ROOT, "optical-flow_amd"))
        import lkflow_amd                                    # the name bench.py looks up is the one rocprofv3 printed
        assert lkflow_amd.lk_kernel_name(rows, cols, win, has_up=True, pairs=pairs) in names
        alg = bench.lk0_bytes_per_pair(cols, rows, levels) * pairs
        assert any(t is not None and alg <= t <= 2 * alg for t in ts)
        assert any(bench.rocprof_avg_launch_ms(k, cols, rows, pairs) for k in names)
    else:
        assert all(t is None for t in ts)                    # measured on other code: not reported
    assert all(bench.measured_traffic(k, cols, rows, pairs + 1) is None for k in names)   # other workloads: null


def test_csrc_digest_follows_the_kernel_sources(tmp_path, monkeypatch):
    import bench
    import __graft_entry__ as ge
    d0 = bench.csrc_digest()
    assert len(d0) == 16 and d0 == bench.csrc_digest() == ge.csrc_digest()
    fake = tmp_path / "repo"
    (fake / "optical-flow_amd" / "csrc").mkdir(parents=True)
    (fake / "include").mkdir()
    (fake / "optical-flow_amd" / "csrc" / "k.h").write_text("a")
    (fake / "include" / "lkflow.h").write_text("h")
    monkeypatch.setattr(ge, "ROOT", str(fake))
    monkeypatch.setattr(ge, "CSRC", str(fake / "optical-flow_amd" / "csrc"))
    d1 = bench.csrc_digest()
    (fake / "optical-flow_amd" / "csrc" / "k.h").write_text("b")
    assert bench.csrc_digest() != d1 and d1 != d0
    # a library says which sources it was built from; one without the stamp (or a missing one) says nothing
    lib = tmp_path / "lib.so"
    lib.write_bytes(b"\x00junk lkflow-mi355x 0.4 (gfx950) src:0123456789abcdef\x00more")
    assert ge.lib_digest(str(lib)) == "0123456789abcdef"
    lib.write_bytes(b"lkflow-mi355x 0.4 (gfx950) src:unstamped")
    assert ge.lib_digest(str(lib)) is None and ge.lib_digest(str(tmp_path / "none.so")) is None
