"""GPU box: why does the predict chain take 52.5 ms inside the bench process and 48.8 ms alone?  Times predict_e2e_bench
(a) in a fresh process, (b) after the bench's training leg (GraphedStep: captured graphs, streams, memory pools alive),
(c) after dropping those objects and emptying the cache."""
import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "--steps", "30", "--warmup", "5", "--skip-cpu-baseline", "--skip-roofline", "--skip-extras"]
import bench
args = bench.parse()
dev = torch.device("cuda:0")
def e2e(tag):
    r = bench.predict_e2e_bench(args, dev)
    st = torch.cuda.memory_stats()
    print(f"{tag}: {r['ms_per_cloud']} ms per cloud; reserved {st['reserved_bytes.all.current'] / 2**30:.2f} GiB", flush=True)
e2e("(a) fresh process")
e2e("(a) again")
res = bench.train_bench(args, dev, 1, 0, 16, 12800, 16, 30, 5)
print("train leg done", flush=True)
e2e("(b) after the training leg")
torch.cuda.empty_cache()
e2e("(b) after empty_cache")
del res; gc.collect(); torch.cuda.empty_cache()
e2e("(c) after gc + empty_cache")
pr = bench.predict_bench(args, dev, reps=2)
print("predict sweep leg:", pr["ms_per_sweep"], flush=True)
e2e("(d) after the predict sweep leg")
torch.cuda.empty_cache()
e2e("(d) after empty_cache")
os.environ["M3D_PREDICT_LOOKAHEAD"] = "1"
e2e("(e) the same with the lookahead")
os.environ["M3D_PREDICT_LOOKAHEAD"] = "0"
e2e("(e) and without again")
