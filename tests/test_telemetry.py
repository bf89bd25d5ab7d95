"""Host-side helpers of the benchmark that need no GPU: the hwmon clock / power sampler (smallhardface_amd/telemetry.py,
bench.py `telemetry`) on a fake sysfs tree, and the launcher parent's GPU count from the KFD topology (bench.py
visible_gpu_count: no HIP call in the process that only starts the ranks)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _fake_card(tmp_path, name, slot, sclk_hz, power_uw, cap_uw=1400000000):
    dev = tmp_path / name / "device"
    hw = dev / "hwmon" / "hwmon3"
    hw.mkdir(parents=True)
    (dev / "vendor").write_text("0x1002\n")
    (dev / "uevent").write_text("DRIVER=amdgpu\nPCI_SLOT_NAME=%s\n" % slot)
    (hw / "freq1_input").write_text("%d\n" % sclk_hz)
    (hw / "power1_input").write_text("%d\n" % power_uw)
    (hw / "power1_cap").write_text("%d\n" % cap_uw)
    return str(hw)


def test_sampler_reads_clock_and_power_of_the_matching_card(tmp_path, monkeypatch):
    from smallhardface_amd import telemetry
    a = _fake_card(tmp_path, "card0", "0000:05:00.0", 2100000000, 900000000)
    b = _fake_card(tmp_path, "card8", "0000:C5:00.0", 1900000000, 1300000000)
    real_glob = telemetry.glob.glob
    monkeypatch.setattr(telemetry.glob, "glob", lambda pat: real_glob(pat.replace("/sys/class/drm", str(tmp_path))))
    assert telemetry.find_card("0000:c5:00.0") == b            # case-insensitive PCI slot match
    assert telemetry.find_card("0000:aa:00.0") is None and telemetry.find_card(None) is None
    assert sorted(telemetry.all_cards()) == sorted([a, b])
    with telemetry.Sampler(b, period_s=0.005) as s:
        time.sleep(0.08)
    t = s.summary()
    assert t["samples"] >= 5 and t["sclk_mhz_mean"] == 1900.0 and t["sclk_mhz_min"] == 1900.0
    assert t["power_w_mean"] == 1300.0 and t["power_cap_w"] == 1400.0 and t["hwmon"] == b and t["cards_sampled"] == 1
    # no card given: every card is sampled, the busiest one (highest mean power) is reported
    with telemetry.Sampler(None, period_s=0.005) as s2:
        time.sleep(0.05)
    t2 = s2.summary()
    assert t2["cards_sampled"] == 2 and t2["hwmon"] == b
    # nothing readable: no summary, no exception
    monkeypatch.setattr(telemetry.glob, "glob", lambda pat: [])
    with telemetry.Sampler(None) as s3:
        pass
    assert s3.summary() is None


def test_visible_gpu_count_from_the_kfd_topology(tmp_path, monkeypatch):
    import glob as _glob
    import bench
    nodes = tmp_path / "kfd" / "kfd" / "topology" / "nodes"
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):       # two CPU nodes, three GPUs
        (nodes / str(i)).mkdir(parents=True)
        (nodes / str(i) / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (64 if simd == 0 else 0, simd))
    real_isdir, real_glob = os.path.isdir, _glob.glob
    monkeypatch.setattr(os.path, "isdir", lambda p: True if p == "/sys/class/kfd" else real_isdir(p))
    monkeypatch.setattr(_glob, "glob", lambda pat, **kw: real_glob(pat.replace("/sys/class/kfd", str(tmp_path / "kfd")), **kw))
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpu_count() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpu_count() == 2
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setattr(os.path, "isdir", lambda p: False if p == "/sys/class/kfd" else real_isdir(p))
    assert bench.visible_gpu_count() == 0                      # no KFD driver at all
