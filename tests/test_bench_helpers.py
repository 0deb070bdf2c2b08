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


def test_replica_plan_one_model_on_n_devices_and_duplicates_on_a_small_box():
    # 8 GPUs: one model, deployed_device_list 0..7 (the reference's arrangement: one cache per device, one server)
    models, rep_model, deployed = bench.plan_replicas(8, 8)
    assert models == ["criteo_dlrm"] and rep_model == ["criteo_dlrm"] * 8 and deployed == {"criteo_dlrm": list(range(8))}
    # 2 replicas, 1 GPU: the second replica is a second deployment of the model on device 0
    models, rep_model, deployed = bench.plan_replicas(2, 1)
    assert models == ["criteo_dlrm", "criteo_dlrm_dup1"] and rep_model == models
    assert deployed == {"criteo_dlrm": [0], "criteo_dlrm_dup1": [0]}
    # 4 replicas, 2 GPUs
    models, rep_model, deployed = bench.plan_replicas(4, 2)
    assert rep_model == ["criteo_dlrm", "criteo_dlrm", "criteo_dlrm_dup1", "criteo_dlrm_dup1"]
    assert deployed == {"criteo_dlrm": [0, 1], "criteo_dlrm_dup1": [0, 1]}
    assert bench.plan_replicas(1, 1) == (["criteo_dlrm"], ["criteo_dlrm"], {"criteo_dlrm": [0]})
