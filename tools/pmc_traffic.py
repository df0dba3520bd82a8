"""Workload for the PMC traffic passes (run under `rocprofv3 --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE`):
two calibration launches that move exactly 4 GiB with 4-byte-per-lane accesses, then two full-OTTO covisitation
builds and two BPR hogwild launches. tools/pmc_summarize.py turns the two CSVs into profiles/<round>/traffic.json."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from otto_amd import _lib
from otto_amd.synth import generate_sessions_torch, OTTO_N_AIDS, OTTO_N_SESSIONS
from otto_amd.covisitation.engine import CovisBuilder
from otto_amd.matrix_factorization.engine import MFEngine, BPR_HOGWILD

dev = torch.device('cuda:0')
lib = _lib.lib()
n = 1 << 30
buf = torch.empty(n, dtype=torch.int32, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.otto_debug_calibrate(C.c_void_p(buf.data_ptr()), n, 1, st)
lib.otto_debug_calibrate(C.c_void_p(buf.data_ptr()), n, 0, st)
torch.cuda.synchronize()
del buf
d = generate_sessions_torch(OTTO_N_SESSIONS, n_aids=OTTO_N_AIDS, seed=42, device=dev)
b = CovisBuilder(OTTO_N_AIDS, kinds=('click_weighted', 'cart_weighted', 'order_weighted'), ts_min=int(d['ts'].min()),
                 ts_max=int(d['ts'].max()), device=dev)
for _ in range(2):
    b.reset()
    b.feed(d['aid'], d['ts'], d['type'], d['sess_off'])
    b.finalize(k=20)
print('stats', b.stats())
del b
rows = 1 << 24
U = torch.randn(OTTO_N_SESSIONS, 64, device=dev) * 0.1
V = torch.randn(OTTO_N_AIDS, 64, device=dev) * 0.1
u = torch.randint(0, OTTO_N_SESSIONS, (rows,), device=dev)
i = torch.randint(0, OTTO_N_AIDS, (rows,), device=dev)
eng = MFEngine(OTTO_N_SESSIONS, OTTO_N_AIDS, 64, rows, device=dev)
for e in range(2):
    eng.bpr_step(U, V, u, i, 42, e, 0, 0.05, 0.0, BPR_HOGWILD)
torch.cuda.synchronize()
del U, V, eng
# reference-config R-MF steps (MSELoss + SparseAdam, 32 factors, batch 262,144)
B, dr = 262144, 32
E1 = torch.randn(OTTO_N_SESSIONS, dr, device=dev)
E2 = torch.randn(OTTO_N_AIDS + 1, dr, device=dev)
st = [torch.zeros_like(E1), torch.zeros_like(E1), torch.zeros_like(E2), torch.zeros_like(E2)]
er = MFEngine(OTTO_N_SESSIONS, OTTO_N_AIDS + 1, dr, B, device=dev)
tg = torch.randint(0, 3, (rows,), device=dev)
lo_ = torch.zeros(1, device=dev)
for b in range(4):
    sl = slice(b * B, (b + 1) * B)
    er.step_sparse_adam(E1, st[0], st[1], E2, st[2], st[3], u[sl], i[sl], tg[sl], 0, 0.05, (0.9, 0.999), 1e-8, b + 1, lo_)
torch.cuda.synchronize()
print('done')
