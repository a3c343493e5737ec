"""Worker of tests/test_distributed_cpu.py::test_tcp_fallback_transport_world3: one rank of the socket-hub transport."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memvul_amd import distributed as d  # noqa: E402

rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
note = d.init_transport(None, rank, world, prefer="tcp", addr="127.0.0.1", port=port)
assert note.startswith("tcp hub")
rows = (np.arange((rank + 2) * 3, dtype=np.float32).reshape(rank + 2, 3) + 100 * rank)
gathered = d.all_gather_rows(rows)
d.barrier()
mx = d.all_reduce_max(float(rank) + 0.5)
s, l = d.all_gather_stats(np.array([0.25 * rank, 0.5], np.float32), np.array([rank & 1, 1], np.uint8))
d.shutdown()
json.dump({"rows": gathered.tolist(), "max": mx, "scores": s.tolist(), "labels": l.tolist()}, open(f"{out}.rank{rank}", "w"))
