"""Host-side helpers of bench.py that have no GPU in them."""
import os

import bench


def test_parse_cpulist():
    assert bench.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert bench.parse_cpulist("") == set()


def _fake_sysfs(root, nodes, numa_node, cpulist):
    for n in range(nodes):
        os.makedirs(root / "devices/system/node" / f"node{n}")
    (root / "devices/system/node" / "possible").write_text("0-1\n")
    d = root / "bus/pci/devices/0000:05:00.0"
    os.makedirs(d)
    (d / "numa_node").write_text(f"{numa_node}\n")
    (d / "local_cpulist").write_text(cpulist)
    return str(root)


def test_gpu_local_cpus_binds_only_on_multi_node_hosts(tmp_path):
    allowed = set(range(0, 128))
    two = _fake_sysfs(tmp_path / "two", 2, 1, "64-127\n")
    assert bench.gpu_local_cpus("0000:05:00.0", allowed, two) == set(range(64, 128))
    # clipped by the affinity mask the rank already has; fewer than 4 CPUs left: no binding
    assert bench.gpu_local_cpus("0000:05:00.0", set(range(60, 70)), two) == set(range(64, 70))
    assert bench.gpu_local_cpus("0000:05:00.0", {1, 2, 64, 65}, two) is None
    one = _fake_sysfs(tmp_path / "one", 1, 0, "0-15\n")
    assert bench.gpu_local_cpus("0000:05:00.0", allowed, one) is None
    unknown = _fake_sysfs(tmp_path / "unk", 2, -1, "0-127\n")
    assert bench.gpu_local_cpus("0000:05:00.0", allowed, unknown) is None
    assert bench.gpu_local_cpus("0000:99:00.0", allowed, two) is None
