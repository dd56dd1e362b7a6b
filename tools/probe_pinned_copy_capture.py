"""Does a small non_blocking H2D copy from pinned memory read its source when it RUNS (after a parked stream is
released) or when it is CALLED?  Decides whether the asynchronous matcher may use hipMemcpyAsync for its matches."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "future-object-detection_amd"))
import torch
from future_od.native import lib as L

dev = torch.device("cuda", 0)
_f = C.c_void_p(); L._plain_call("fod_host_flag_create", C.addressof(_f)); flag = _f.value
ticket = 0
for words in (64, 256, 1024, 1536, 4096, 65536):
    ticket += 1
    h = torch.zeros(words, dtype=torch.int32).pin_memory()
    torch.cuda.synchronize()
    s = torch.cuda.current_stream(dev).cuda_stream
    L._plain_call("fod_stream_wait_flag", flag, ticket, s)
    d = h.to(dev, non_blocking=True)            # queued behind the parked wait
    time.sleep(0.02)
    h.fill_(7)                                   # the "worker" writes its result afterwards
    L._plain_call("fod_host_flag_set", flag, ticket)
    torch.cuda.synchronize()
    got = int((d == 7).sum())
    print(f"{words * 4:7d} bytes: {got}/{words} words carry the value written AFTER the call "
          f"({'read at run time' if got == words else 'captured at call time (stale)'})")
