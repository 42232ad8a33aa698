"""TEST INFRASTRUCTURE: an mcq_exchange_fn that moves the blocks of the sharded path through the host and
torch.distributed (gloo), for rehearsing n_ranks > 1 on a box with one GPU (RCCL refuses two ranks on one device)."""
import ctypes as C
import importlib
import traceback

import torch
import torch.distributed as dist


def make_gloo_exchange(group=None):
    eng = importlib.import_module("metacache-mpi_amd.engine")
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int

    def fn(user, send_base, send_off, send_bytes, recv_base, recv_off, recv_bytes, n, rank):
        try:
            sb = [int(send_bytes[p]) for p in range(n)]
            rb = [int(recv_bytes[p]) for p in range(n)]
            send = torch.empty(max(1, sum(sb)), dtype=torch.uint8)
            pos = 0
            for p in range(n):
                if sb[p]:
                    if hip.hipMemcpy(send.data_ptr() + pos, (send_base or 0) + int(send_off[p]), sb[p], 2) != 0:   # device -> host
                        return 1
                    pos += sb[p]
            recv = torch.empty(max(1, sum(rb)), dtype=torch.uint8)
            dist.all_to_all_single(recv[:sum(rb)], send[:sum(sb)], rb, sb, group=group)
            pos = 0
            for p in range(n):
                if rb[p]:
                    if hip.hipMemcpy((recv_base or 0) + int(recv_off[p]), recv.data_ptr() + pos, rb[p], 1) != 0:   # host -> device
                        return 1
                    pos += rb[p]
            return 0
        except Exception:
            traceback.print_exc()
            return 1
    return eng.EXCHANGE_FN(fn)
