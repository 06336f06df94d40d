"""IndexIVFFlat build on the device with the library's own kernels (csrc/ivf_build.hip): train (spherical k-means) and add + grouping of
1M x 512 rows into 3162 lists, and that two trainings give the same bits:  python tools/ivf_build_time.py"""
import sys, time, torch
sys.path.insert(0, ".")
from wise_amd.index.ivf_flat import IVFFlatIPIndex
n, d, nlist = 1_000_000, 512, 3162
x = torch.nn.functional.normalize(torch.randn(n, d, device="cuda"), dim=1)
idx = IVFFlatIPIndex(d, nlist)
torch.cuda.synchronize(); t0 = time.perf_counter()
idx.train(x); torch.cuda.synchronize(); t1 = time.perf_counter()
idx.add_with_ids(x, torch.arange(n)); idx._finalize(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"train {t1 - t0:.2f} s ({idx.niter} iterations), add + finalize {t2 - t1:.2f} s")
c2 = IVFFlatIPIndex(d, nlist); c2.train(x)
print("deterministic:", bool(torch.equal(idx.centroids, c2.centroids)))
