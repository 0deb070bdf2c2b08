"""Stand-alone run of bench.py's single-entry config-3 leg (tools: where does a round of concurrent instances go?).
    python tools/entry_probe.py [P] [rows_log2] [steps]"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    rl = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    sys.argv = sys.argv[:1]
    import faulthandler
    faulthandler.dump_traceback_later(int(__import__("os").environ.get("ENTRY_PROBE_DUMP_S", "120")), exit=True)
    a = bench.parse_args()
    a.batch = int(__import__("os").environ.get("ENTRY_PROBE_BATCH", a.batch))
    import torch
    from hugectr_backend_amd import hps
    ndev = torch.cuda.device_count()
    res = bench.c3_single_entry_leg(a, torch, hps, [g % ndev for g in range(P)], 1 << rl, steps)
    print(json.dumps(bench._sig(res), indent=1))


if __name__ == "__main__":
    main()
