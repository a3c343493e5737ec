"""Round 6: the drop-in END TO END on the shipped library — host work included — in the three forms of
memvul_amd.predict_memory.test_siamese (reference: predict_memory.py:49-114):
  instances   the literal drop-in under test_config_memory.json: reader -> Instances -> DataLoader (pad-to-longest collation) -> model(**batch) per batch ->
              make_output_human_readable -> one JSON line per batch (predict_memory.py:92-110, model_memory.py:118-191)
  sweep       sweep=True: the same Instances scored in one resident length-bucketed sweep
  arrays      sweep="arrays": batched tokenisation straight into arrays, records written by a thread while the engine runs
on N ragged synthetic issue reports (precise compute dtype, 124 anchors, batch 512), tokenised by the REAL BertTokenizerFast over a synthetic 30 522-entry
WordPiece vocabulary (there is no bert-base-uncased vocab.txt on disk: the vocabulary here has the special tokens at BERT's ids, a dictionary of whole words
and the ## pieces that spell everything else — the tokenizer's work per character is the real one).  Every stage that can bound a form is timed inside it.
Usage (GPU box): python scripts/r06_e2e_dropin.py [N]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def make_vocab(path, rng, n=30522):
    """bert-base-uncased's layout: [PAD] 0, [unused*], [UNK] 100, [CLS] 101, [SEP] 102, [MASK] 103, single characters, whole words, ## pieces."""
    import string

    toks = ["[PAD]"] + ["[unused%d]" % i for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + ["[unused%d]" % i for i in range(99, 994)]
    chars = list(string.ascii_lowercase + string.digits + string.punctuation)
    toks += chars + ["##" + c for c in string.ascii_lowercase + string.digits]
    import plumbing_util as pu
    words = set(pu.WORDS)
    syll = ["ab", "ac", "ad", "al", "an", "ar", "as", "at", "ba", "be", "bu", "ca", "co", "cr", "da", "de", "di", "do", "en", "er", "es", "ex", "fi", "fo", "ge", "ha",
            "he", "in", "io", "is", "it", "le", "li", "lo", "ma", "me", "mo", "ne", "no", "nu", "of", "on", "or", "ou", "pa", "pe", "po", "pr", "re", "ro", "sa", "se",
            "si", "so", "st", "ta", "te", "th", "ti", "to", "tr", "un", "ur", "us", "ve", "wa", "we", "wi"]
    while len(words) < 14000:
        words.add("".join(syll[j] for j in rng.integers(0, len(syll), size=int(rng.integers(2, 5)))))
    words = sorted(words)
    pieces = set()
    while len(toks) + len(words) + len(pieces) < n:
        pieces.add("##" + "".join(syll[j] for j in rng.integers(0, len(syll), size=int(rng.integers(1, 4)))))  # (1 - 3 syllables: 68 + 68^2 + 68^3 distinct pieces)
    toks += words + sorted(pieces)
    toks = toks[:n]
    assert len(set(toks)) == len(toks) and toks[101] == "[CLS]" and toks[102] == "[SEP]", (len(set(toks)), len(toks))
    open(path, "w", encoding="utf-8").write("\n".join(toks) + "\n")
    return words


def make_corpus(root, rng, words, n_irs, n_anchors=124):
    """Issue reports of realistic shape: a title and a body of dictionary words, identifiers that are NOT in the vocabulary (split into pieces), numbers,
    punctuation; body lengths such that the token lengths spread over ~20 .. 256 with a tail that is truncated at 256."""
    words = list(words)
    letters = np.array(list("abcdefghijklmnopqrstuvwxyz_"))
    idents = ["".join(letters[rng.integers(0, len(letters), size=int(k))]) for k in rng.integers(5, 14, size=4096)]  # not in the vocabulary: split into pieces
    punct = list(".,:;()[]/-")

    def text(nw):
        r = rng.random(nw)
        wi = rng.integers(0, len(words), size=nw)
        ii = rng.integers(0, len(idents), size=nw)
        num = rng.integers(0, 100000, size=nw)
        pi = rng.integers(0, len(punct), size=nw)
        return " ".join(words[wi[j]] if r[j] < 0.8 else idents[ii[j]] if r[j] < 0.92 else str(num[j]) if r[j] < 0.96 else punct[pi[j]] for j in range(nw))

    cwes = ["CWE-%d" % (100 + i) for i in range(n_anchors)]
    golden = os.path.join(root, "CWE_anchor_golden_project.json")
    json.dump({c: text(int(rng.integers(30, 260))) for c in cwes}, open(golden, "w"))
    recs = []
    for i in range(n_irs):
        pos = i % 311 == 7
        recs.append({"Issue_Title": text(int(rng.integers(4, 12))), "Issue_Body": text(int(rng.integers(8, 200))),
                     "Security_Issue_Full": "1" if pos else "0", "Issue_Url": "https://example.invalid/issues/%d" % i,
                     "CVE_ID": "CVE-2020-%d" % i if pos else None, "CWE_ID": cwes[int(rng.integers(0, len(cwes)))] if pos else None})
    test_path = os.path.join(root, "test_project.json")
    json.dump(recs, open(test_path, "w"))
    return golden, test_path


class Stage:
    """Wall-clock accumulators around callables (a stage's time includes whatever it calls)."""

    def __init__(self):
        self.t = {}

    def wrap(self, obj, name, label):
        f = getattr(obj, name)
        acc = self.t

        def g(*a, **k):
            t0 = time.perf_counter()
            try:
                return f(*a, **k)
            finally:
                acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0

        setattr(obj, name, g)
        return f


def main():
    if os.environ.get("MEMVUL_PROBE_SWITCH_INTERVAL"):  # A/B of the interpreter's thread switch interval (default 5 ms)
        sys.setswitchinterval(float(os.environ["MEMVUL_PROBE_SWITCH_INTERVAL"]))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
    import plumbing_util as pu
    from memvul_amd import data as mvdata
    from memvul_amd import predict_memory as pm

    rng = np.random.default_rng(11)
    root, arch, _, _, w, dims = pu.make_fixture(n_irs=4, n_anchors=4, layers=12)
    vocab = os.path.join(root, "vocab.txt")
    words = make_vocab(vocab, rng)
    os.environ["MEMVUL_BERT_VOCAB"] = vocab
    os.environ.pop("MEMVUL_ALLOW_HASH_TOKENIZER", None)
    golden, test_path = make_corpus(root, rng, words, n)
    out = os.path.join(root, "test_results", "pred.json")
    eo = dict(max_tokens=512 * 256, max_batch=512, max_anchors=128)
    print("N = %d issue reports, 124 anchors, batch 512, precise compute dtype (the default), host cores %d, real BertTokenizerFast over a synthetic %d-entry "
          "WordPiece vocabulary" % (n, os.cpu_count(), 30522), flush=True)
    results = {}
    os.environ["MEMVUL_RECORD_WORKERS"] = sys.argv[2] if len(sys.argv) > 2 else "0"
    print("MEMVUL_RECORD_WORKERS = %s (processes formatting the JSON records; 0 = in the driver process)" % os.environ["MEMVUL_RECORD_WORKERS"])
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
    for form, kw in (("arrays", dict(sweep="arrays")), ("sweep", dict(sweep=True)), ("instances", dict(sweep=False))):
        if only and form not in only:
            continue
        st = Stage()
        import memvul_amd.reader_memory as rm
        import memvul_amd.model_memory as mm
        import memvul_amd.archive as ar
        undo = []
        undo.append((rm.ReaderMemory, "read_arrays", st.wrap(rm.ReaderMemory, "read_arrays", "reader: read_arrays (JSON + batched WordPiece)")))
        undo.append((rm.ReaderMemory, "read_dataset", st.wrap(rm.ReaderMemory, "read_dataset", "reader: JSON + WordPiece tokenisation (read_dataset)")))
        undo.append((mm.ModelMemory, "make_output_human_readable", st.wrap(mm.ModelMemory, "make_output_human_readable", "make_output_human_readable (records as dicts)")))
        undo.append((mm.ModelMemory, "sweep_arrays", st.wrap(mm.ModelMemory, "sweep_arrays", "engine: sweep_arrays (upload + bucketed sweep + download)")))
        undo.append((mm.ModelMemory, "forward_on_instances", st.wrap(mm.ModelMemory, "forward_on_instances", "anchor bank (forward_on_instances x 1)")))
        undo.append((mvdata, "collate", st.wrap(mvdata, "collate", "collate (pad-to-longest, Instances -> arrays)")))
        import memvul_amd.binding as mb
        import memvul_amd.records as mr
        undo.append((mb.Engine, "forward_by_length", st.wrap(mb.Engine, "forward_by_length", "scorer thread: engine.forward_by_length (sort + upload + passes + download)")))
        undo.append((mm.ModelMemory, "__call__", st.wrap(mm.ModelMemory, "__call__", "scorer thread: model(**batch) in all")))
        undo.append((mr.RecordWriter, "submit", st.wrap(mr.RecordWriter, "submit", "writer: RecordWriter.submit (format + write)")))
        import memvul_amd.custom_metric as cmm
        undo.append((cmm.SiameseMeasureV1, "__call__", st.wrap(cmm.SiameseMeasureV1, "__call__", "scorer thread: _siamese_metric")))
        undo.append((mm._ClassificationCounts, "__call__", st.wrap(mm._ClassificationCounts, "__call__", "scorer thread: _counts")))
        _il = mm.ModelMemory.__dict__["_ids_lens"].__func__
        acc_il = st.t

        def _il_timed(sample, _f=_il):
            t0 = time.perf_counter()
            try:
                return _f(sample)
            finally:
                acc_il["scorer thread: _ids_lens"] = acc_il.get("scorer thread: _ids_lens", 0.0) + time.perf_counter() - t0

        mm.ModelMemory._ids_lens = staticmethod(_il_timed)
        undo.append((mm.ModelMemory, "_ids_lens", staticmethod(_il)))
        undo.append((pm, "load_archive", st.wrap(pm, "load_archive", "load_archive (weights -> engine)")))
        undo.append((json, "dumps", st.wrap(json, "dumps", "json.dumps of the records")))
        t0 = time.perf_counter()
        m = pm.test_siamese(arch, test_path, golden, test_config=pu.TEST_CONFIG, predictions_output_file=out, batch_size=512, engine_options=eo, **kw)
        total = time.perf_counter() - t0
        for o, name, f in undo:
            setattr(o, name, f)
        fixed = st.t.get("load_archive (weights -> engine)", 0) + st.t.get("anchor bank (forward_on_instances x 1)", 0)
        results[form] = dict(total_s=total, job_s=total - fixed, irs_per_s=n / (total - fixed), stages=st.t, s_f1=m["s_f1-score"], bytes=os.path.getsize(out))
        print("\n== test_siamese(%s): %.2f s in all; without the one-off archive load + anchor bank %.2f s = %.0f issue reports/s whole job (read + tokenise + score + "
              "records + metrics); predictions file %.1f MB, s_f1 %.6f" % (", ".join("%s=%r" % kv for kv in kw.items()), total, total - fixed, n / (total - fixed),
                                                                           os.path.getsize(out) / 1e6, m["s_f1-score"]), flush=True)
        for k, v in sorted(st.t.items(), key=lambda kv: -kv[1]):
            print("   %-62s %7.2f s  (%.0f IR/s if it were alone)" % (k, v, n / v if v > 0 else 0))
    print("\n" + json.dumps({k: {kk: vv for kk, vv in v.items()} for k, v in results.items()}))


if __name__ == "__main__":
    main()
