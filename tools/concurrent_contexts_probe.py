import sys, importlib, json, numpy as np, threading
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
pkg = importlib.import_module("monte-carlo-ray-tracer_amd")
man = json.load(open('/root/repo/tests/golden/manifest.json'))
import test_film_filters as T
img, cam, r = T._case(pkg, man, "film_mitchell")
n = 2
ctxs = [pkg.Context(0) for _ in range(n)]
for c in ctxs: c.upload_image(img)
def shard(i):
    sh = cam.copy(); sh.shard_count, sh.shard_index, sh.shard_rows = n, i, 8
    return sh
seq = []
for i in range(n):
    b = torch.full((cam.height, cam.width, 4), float("nan"), dtype=torch.float64, device="cuda:0")
    ctxs[i].render_film_device(shard(i), man["seed"], pkg.INTEGRATOR_PATH_TRACER, b.data_ptr()); st = ctxs[i].render_finish()
    seq.append((b.cpu().numpy(), st))
for rep in range(3):
    bufs = [torch.full((cam.height, cam.width, 4), float("nan"), dtype=torch.float64, device="cuda:0") for _ in range(n)]
    stats = [None] * n
    def work(i):
        ctxs[i].render_film_device(shard(i), man["seed"], pkg.INTEGRATOR_PATH_TRACER, bufs[i].data_ptr()); stats[i] = ctxs[i].render_finish()
    th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in th: t.start()
    for t in th: t.join()
    for i in range(n):
        a = bufs[i].cpu().numpy(); b = seq[i][0]
        d = a - b
        print("rep", rep, "ctx", i, "sum conc", a.sum(axis=(0, 1)), "seq", b.sum(axis=(0, 1)), "maxabs diff", np.abs(d).max(), "n diff", int((d != 0).sum()),
              "rays", stats[i]["rays"], seq[i][1]["rays"], "iters", stats[i].get("launches"), seq[i][1].get("launches"), flush=True)
