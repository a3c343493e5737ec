"""Round 6: the host-side ceiling of the Instance form of the drop-in (test_siamese(sweep=False): reader -> streamed WordPiece tokenisation -> Instances -> DataLoader ->
model(**batch) -> records): the engine replaced by a stand-in that answers at a given rate (0 = at once), everything else the product's code.
Usage: python scripts/r06_host_bound_probe.py [N] [engine IR/s, 0 = infinite] [record workers]"""
import os, sys, time, json, threading
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests'); sys.path.insert(0, ROOT+'/scripts')
import numpy as np
import r06_e2e_dropin as E
import plumbing_util as pu
from memvul_amd import data as mvdata, predict_memory as pm, model_memory as mm, records as R
import memvul_amd.binding as binding

class FakeEngine:
    BY_LENGTH_MIN_TOKENS = binding.Engine.BY_LENGTH_MIN_TOKENS
    forward_by_length = binding.Engine.forward_by_length
    def __init__(self, device=0, **kw): self.P = kw.get("proj_dim", 512); self.G = 0
    def load_state_dict(self, sd, cd=1): pass
    def close(self): pass
    def anchor_reset(self): self.G = 0
    def anchor_append(self, ids, lens): self.G += ids.shape[0]
    @property
    def n_anchors(self): return self.G
    def anchor_get(self): return np.zeros((self.G, self.P), np.float32)
    def forward(self, ids, lens, want_logits=True, want_probs=True, want_embed=False):
        B = ids.shape[0]
        if RATE > 0:
            time.sleep(B / RATE)   # the GPU at RATE issue reports/s, GIL released
        p = np.random.default_rng(B).random((B, self.G, 2), dtype=np.float32)
        return {"logits": None, "probs": p, "best": p[:, 0], "best_idx": np.zeros(B, np.int32), "embed": None}
mm.Engine = FakeEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
RATE = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
os.environ["MEMVUL_RECORD_WORKERS"] = sys.argv[3] if len(sys.argv) > 3 else "0"
rng = np.random.default_rng(11)
root, arch, _, _, w, dims = pu.make_fixture(n_irs=4, n_anchors=4, layers=12)
vocab = os.path.join(root, "vocab.txt"); words = E.make_vocab(vocab, rng)
os.environ["MEMVUL_BERT_VOCAB"] = vocab
golden, test_path = E.make_corpus(root, rng, words, n)
out = os.path.join(root, "test_results", "pred.json")
acc = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try: return f(*a, **k)
        finally: acc[label] = acc.get(label, 0) + time.perf_counter() - t0
    setattr(obj, name, g)
wrap(mm.ModelMemory, "forward", "scorer: ModelMemory.forward (all)")

wrap(FakeEngine, "forward", "scorer: engine (sleep)")
wrap(mm.ModelMemory, "make_output_human_readable", "scorer?: make_output_human_readable")
wrap(R.RecordWriter, "submit", "main: RecordWriter.submit")
wrap(mvdata, "collate", "collator: collate")
wrap(pm, "evaluate", "evaluate (whole loop)")
wrap(pm, "load_archive", "load_archive")
import memvul_amd.reader_memory as rm
wrap(rm.ReaderMemory, "read_dataset", "read_dataset")
wrap(mm.ModelMemory, "__call__", "scorer: model(**batch)")
wrap(mm.ModelMemory, "_siamese_metric", "scorer: _siamese_metric") if hasattr(mm.ModelMemory, "_siamese_metric") else None
t0 = time.perf_counter()
m = pm.test_siamese(arch, test_path, golden, test_config=pu.TEST_CONFIG, predictions_output_file=out, batch_size=512, engine_options={}, sweep=False)
tot = time.perf_counter() - t0
job = acc.get("evaluate (whole loop)", tot)
print("N %d, engine stand-in at %s IR/s, record workers %s: evaluate loop %.2f s = %.0f IR/s (total with archive load %.2f s)" % (n, ("%.0f" % RATE) if RATE > 0 else "infinite", os.environ["MEMVUL_RECORD_WORKERS"], job, n / job, tot))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print("  %-50s %.2f s" % (k, v))
