"""CPU: the host logic of bench.py that needs no GPU -- its configurations, the roofline arithmetic against the committed
per-ray records, and the child-process harness of the additional configurations (a child that fails or stalls must become an
error entry, not a missing headline line)."""
import json
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_configurations_and_records():
    cfgs = bench.configurations()
    assert list(cfgs)[0] == "cornell" and set(cfgs) == {"cornell", "veach-mis", "dragon-class", "dragon-partial", "material-zoo", "teapot-class"}
    for name, (label, make, spp, fpl) in cfgs.items():
        assert spp == fpl and callable(make) and str(spp) in label  # one launch per job
        rec = bench.pmc_per_ray(name)
        if name in ("dragon-partial", "material-zoo"):  # timed only: no PMC passes of their own
            continue
        assert rec and rec["valu_wave_insts_per_ray"] > 0 and os.path.exists(os.path.join(ROOT, rec["source"])), name


def test_pmc_staleness_is_reported():
    """the per-ray instruction / byte counts come from committed profiles; bench.py says when the kernel sources
    have changed since (kernel_source_hash stamped by tools/summarize_profiles.py)."""
    h = bench.kernel_source_hash()
    assert len(h) == 16 and h == bench.kernel_source_hash()
    rl = bench.rooflines("cornell", 1.0e11, 256, 500.0)
    rec = bench.pmc_per_ray("cornell")
    assert rl["valu"]["pmc_stale"] == (rec.get("kernel_source_hash") != h)
    assert 0.0 < rl["valu"]["useful_lane_frac"] <= rl["valu"]["frac"]


def test_roofline_fractions_are_fractions():
    for name in bench.configurations():
        if name in ("dragon-partial", "material-zoo"):
            continue
        rl = bench.rooflines(name, 1.0e10 if name != "cornell" else 1.2e11, 256, 500.0)
        assert 0.0 < rl["valu"]["frac"] <= 1.0 and rl["valu"]["peak"] == pytest.approx(78.6432)
        assert 0.0 <= rl["hbm"]["frac"] <= 1.0 and rl["hbm"]["peak"] == 8000.0


def test_a_failing_or_stalling_configuration_becomes_an_error_entry(monkeypatch, tmp_path):
    """configs_in_children runs `bench.py --config-child NAME` per configuration; here the 'bench.py' it starts is a stand-in
    that succeeds for one name, fails for another and sleeps past the timeout for the third."""
    fake = tmp_path / "fake_bench.py"
    fake.write_text(
        "import json, sys, time\n"
        "name = sys.argv[sys.argv.index('--config-child') + 1]\n"
        "if name == 'veach-mis': print(json.dumps({'config': {'value': 1.0}}))\n"
        "elif name == 'dragon-class': sys.stderr.write('boom'); sys.exit(3)\n"
        "else: time.sleep(30)\n")
    monkeypatch.setattr(bench, "__file__", str(fake))
    out = bench.configs_in_children("cornell", timeout_s=2.0)
    assert out["veach-mis"] == {"value": 1.0}
    assert "exit code 3" in out["dragon-class"]["error"] and "boom" in out["dragon-class"]["stderr_tail"]
    assert "did not finish" in out["teapot-class"]["error"] and "did not finish" in out["dragon-partial"]["error"] and "did not finish" in out["material-zoo"]["error"]
    json.dumps(out)


def test_the_committed_bench_line_carries_every_configuration():
    """bench.py's `configs` (the latest committed line under profiles/): every BASELINE configuration beside the headline one, each rendered by
    the library's default path -- since round 4 eight frame chains per pixel on every kernel (`frame_chains`); round 3's lines timed the BVH
    configurations twice (strict frame order and two opt-in chains)."""
    import glob
    path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench.json")))[-1]
    d = json.loads(open(path).read().strip().splitlines()[-1])
    assert d["roofline"]["pmc_stale"] is False, path
    for name in ("veach-mis", "dragon-class", "teapot-class", "dragon-partial", "material-zoo"):
        c = d["configs"][name]
        assert c["value"] > 0 and c["rays"] > 0, name
        if "frame_chains" in c:
            assert c["frame_chains"] == 8 and "strict_order" not in c, name
        else:  # a round-3 line
            assert c["frame_groups"] in (1, 2), name


def test_the_watchdog_turns_a_hung_phase_into_an_error_record_and_a_non_zero_exit():
    """`bench.py --gpus N` must not hang the driver's one scaling run.  A phase that overruns (here: a sleep standing in for
    a rendezvous that never completes) makes rank 0 print ONE JSON error record -- same leading keys as the bench line, value null -- and the
    process leaves with exit code 3, from the watchdog's thread (the main thread never returns); a phase that ends in time leaves no trace."""
    import subprocess
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "import bench\n"
            "w = bench.Watchdog(0, 2, 2)\n"
            "w.arm('quick phase', 5.0); w.disarm()\n"
            "w.arm('rendezvous that never completes', 0.5)\n"
            "time.sleep(30)\n"
            "print('NOT REACHED')\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=25)
    assert p.returncode == 3, (p.returncode, p.stderr[-300:])
    assert "NOT REACHED" not in p.stdout
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    assert rec["value"] is None and rec["n_gpus"] == 2 and "rendezvous that never completes" in rec["error"] and rec["metric"] == "Mrays/s"
    # a rank other than 0 leaves as loudly, without a record
    p = subprocess.run([sys.executable, "-c", code.replace("Watchdog(0, 2, 2)", "Watchdog(1, 2, 2)")], capture_output=True, text=True, timeout=25)
    assert p.returncode == 3 and p.stdout.strip() == "" and "giving up" in p.stderr
