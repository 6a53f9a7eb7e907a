import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["POA_WS_PROF"] = "1"
from poasta_amd import aligner, workloads as W
g, (qseq, qoff) = W.config2(n_queries=10000)
rb = aligner.ResidentBatch(g, qseq, qoff)
rb.run(aligner.GapAffine(4, 2, 6), None, aligner.make_config("hybrid", queue_entries_per_cell=0.25))
sc = rb.search_counters()
st = rb.stats()
print("ms_exact", st["ms_exact"], "steps mean", sc[:, 3].mean())
