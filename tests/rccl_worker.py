"""Worker of tests/test_rccl_two_ranks_gpu.py: ONE rank of a real N-rank RCCL communicator (one process per GPU, no torch).
argv: rank world port out_prefix.  Everything bench.py --gpus N and test_siamese_sharded do with the transport, on small data:
agreement over the rendezvous hub -> ncclCommInitRank inside libmemvul_hip.so -> the (score, label) all-gather on the engine's
stream next to engine work."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memvul_amd import distributed as d  # noqa: E402
from memvul_amd import synth  # noqa: E402
from memvul_amd.binding import Engine  # noqa: E402

rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
dims = synth.BertDims(layers=1, vocab_size=1024)
eng = Engine(rank, vocab_size=1024, layers=1, max_tokens=2048, max_batch=8, max_anchors=8)  # device = rank: one GPU per process
eng.load_state_dict(synth.make_weights(dims))
note = d.init_transport(eng, rank, world, prefer="rccl", addr="127.0.0.1", port=port)
info = eng.comm_info()
ids, lens = synth.make_ids(4, 64, 1024, seed=synth.SEED + rank)
u = eng.encode(ids, lens)                                 # engine work and the collective share the stream
n = 3 + 2 * rank                                           # ragged blocks: the count gather + zero-padded block path
rng = np.random.default_rng(100 + rank)
scores = rng.random(n, dtype=np.float32)
labels = (rng.random(n) < 0.3).astype(np.uint8)
s, l = d.all_gather_stats(scores, labels)
rows = d.all_gather_rows(u[:, :8].astype(np.float32))
d.barrier()
mx = d.all_reduce_max(float(rank) + 0.5)
d.shutdown()
eng.close()
json.dump({"note": note, "info": info, "scores": s.tolist(), "labels": l.tolist(), "rows": rows.tolist(), "own_rows": u[:, :8].tolist(), "max": mx},
          open(f"{out}.rank{rank}", "w"))
