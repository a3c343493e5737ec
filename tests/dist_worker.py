"""world_size-2 worker (gloo on CPU): shard a synthetic corpus' (score,label) stats, exchange them with the
single all-gather of the multi-GPU path, and check the result equals the unsharded arrays bit-for-bit."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from memvul_amd import custom_metric as cm  # noqa: E402
from memvul_amd import distributed as mvdist  # noqa: E402
from memvul_amd import synth  # noqa: E402


def main():
    out_path = sys.argv[1]
    n = int(sys.argv[2])
    dist = mvdist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(123)
    labels = synth.make_labels(n, pos_rate=0.1)
    scores = np.clip(rng.normal(0.5 + 0.2 * labels, 0.15), 0, 1).astype(np.float32)
    first, count = mvdist.shard_range(n, rank, world)
    s, l = mvdist.all_gather_stats(scores[first:first + count], labels[first:first + count])
    ok = bool(np.array_equal(s, scores) and np.array_equal(l, labels))
    # threshold table is additive: all-reduce alternative gives the same best threshold
    import torch
    t = torch.from_numpy(cm.threshold_confusion_table(labels[first:first + count], scores[first:first + count]))
    dist.all_reduce(t)
    same_best = cm.best_from_table(t.numpy()) == cm.find_best_thres(labels, scores)
    m = cm.siamese_metrics(l, s)
    mvdist.barrier()
    tmax = mvdist.all_reduce_max(float(rank))
    if rank == 0:
        json.dump({"ok": ok, "same_best": bool(same_best), "world": world, "tmax": tmax, "f1": m["f1"], "thres": float(m["thres"]),
                   "auc": float(m["auc"]), "ap": float(m["ave_precision_score"])}, open(out_path, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
