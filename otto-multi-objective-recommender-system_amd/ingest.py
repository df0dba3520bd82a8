"""Host -> device event ingest: pinned staging, a copy stream, and the next chunk's transfer
overlapped with the expansion of the current one.

The C-ABI takes device pointers; the reference's scripts start from host frames (``pd.read_parquet`` /
``pd.read_pickle``, ``src/covisitation/inference.py:87``, ``src/ranker/aid_feature_engineering.py:21-36``). This module
is the piece in between for the builder script: session chunks of the host SoA (``events.frame_to_events``) are copied
into page-locked staging buffers, sent with asynchronous copies on a side stream and handed to ``CovisBuilder.feed``
on the compute stream; two staging sets alternate, so chunk i + 1 crosses PCIe while chunk i is expanded.
Results do not depend on the chunking (tests/test_covis_gpu.py::test_chunked_feed_equals_single_feed).
"""
import numpy as np


def feed_host_events(builder, ev, device, chunk_sessions=2_000_000, pinned=True):
    """Feed the host :class:`synth.Events` ``ev`` to ``builder`` (a ``CovisBuilder`` on ``device``) chunk by chunk.
    Returns (seconds spent in this call, bytes sent)."""
    import time
    import torch
    dev = torch.device(device)
    S = ev.n_sessions
    if S == 0:
        return 0.0, 0
    bounds = list(range(0, S, chunk_sessions)) + [S]
    max_e = max(int(ev.sess_off[bounds[i + 1]] - ev.sess_off[bounds[i]]) for i in range(len(bounds) - 1))
    max_s = max(bounds[i + 1] - bounds[i] for i in range(len(bounds) - 1))

    def staging():
        mk = lambda n, dt: torch.empty(n, dtype=dt, pin_memory=pinned)
        return mk(max_e, torch.int32), mk(max_e, torch.int32), mk(max_e, torch.uint8), mk(max_s + 1, torch.int64)
    host = [staging(), staging()]
    devb = [tuple(torch.empty_like(t, device=dev) for t in h) for h in host]
    copy_stream = torch.cuda.Stream(device=dev)
    compute = torch.cuda.current_stream(dev)
    sent = [torch.cuda.Event(), torch.cuda.Event()]          # H2D of the set finished
    used = [torch.cuda.Event(), torch.cuda.Event()]          # feed() no longer reads the set
    aid_i32 = ev.aid.view(np.int32) if ev.aid.dtype == np.uint32 else ev.aid.astype(np.int32)
    t0 = time.time()
    nbytes = 0
    for c in range(len(bounds) - 1):
        b = c & 1
        lo, hi = bounds[c], bounds[c + 1]
        e0, e1 = int(ev.sess_off[lo]), int(ev.sess_off[hi])
        ne, ns = e1 - e0, hi - lo
        if c >= 2:
            used[b].synchronize()                            # the staging set is free again (host and device side)
        h, d = host[b], devb[b]
        # plain NumPy slice copies into the page-locked set (measured faster here than torch's threaded CPU copy)
        h[0][:ne].numpy()[...] = aid_i32[e0:e1]
        h[1][:ne].numpy()[...] = ev.ts[e0:e1]
        h[2][:ne].numpy()[...] = ev.type[e0:e1]
        h[3][:ns + 1].numpy()[...] = ev.sess_off[lo:hi + 1] - e0
        with torch.cuda.stream(copy_stream):
            for k, n in ((0, ne), (1, ne), (2, ne), (3, ns + 1)):
                d[k][:n].copy_(h[k][:n], non_blocking=True)
            sent[b].record(copy_stream)
        nbytes += 9 * ne + 8 * (ns + 1)
        compute.wait_event(sent[b])
        builder.feed(d[0][:ne], d[1][:ne], d[2][:ne], d[3][:ns + 1])
        used[b].record(compute)
    torch.cuda.synchronize(dev)
    return time.time() - t0, nbytes
