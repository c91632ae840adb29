export TMPDIR=/tmp
TRLDA_BATCH_TRACE=1 python3 - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -12
import time, sys
sys.path.insert(0, '.')
from trlda_amd.documents import CSRDocuments, DeviceBatch
from trlda_amd.utils.synthetic import make_corpus
from trlda_amd import _ffi
L=_ffi.lib()
docs = CSRDocuments(*make_corpus(200, 7000, seed=20150707, mean_unique=100))
for _ in range(20): DeviceBatch(docs, 7000, 0).close()
t=time.perf_counter()
for _ in range(200): DeviceBatch(docs, 7000, 0).close()
print("create+close %.1f us" % ((time.perf_counter()-t)/200*1e6))
hs=[]
t=time.perf_counter()
for _ in range(200): hs.append(DeviceBatch(docs, 7000, 0))
t1=time.perf_counter()
for h in hs: h.close()
t2=time.perf_counter()
print("create %.1f us  close %.1f us" % ((t1-t)/200*1e6,(t2-t1)/200*1e6))
import ctypes as C
h=_ffi.vp()
t=time.perf_counter()
for _ in range(200):
    L.trlda_batch_create(C.byref(h), 0, 7000, 200, docs.indptr, docs.ids, docs.cnts); L.trlda_batch_destroy(h)
print("raw ctypes create+destroy %.1f us" % ((time.perf_counter()-t)/200*1e6))
PY
