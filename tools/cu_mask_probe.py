#!/usr/bin/env python3
"""Do CU-masked streams (hipExtStreamCreateWithCUMask) let a chain of tiny kernels (the eigensolver) run NEXT TO
chip-filling convolutions?  Times a conv chain and an eigen chain alone and together, for several mask layouts.

Outcome on the MI355X box (round 2): two PLAIN streams already overlap the two chains (13.6 ms + 5.7 ms alone, 14.9 ms
together); a 16-CU mask in the low bits sits in one XCD and slows the convolutions by a third (18.3 ms) without
helping; the second pair of masked streams could not be created (hipExtStreamCreateWithCUMask -> 2: the process's four
hardware queues were taken).  CU masks are not used by the library; the lanes of streams.py are."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from style_transfer_amd import eigh, ops  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(s.value)


def timed(fn_by_stream):
    """{stream: fn}: start all, wait for all; returns per-stream elapsed ms and the wall time."""
    torch.cuda.synchronize()
    ev = {}
    t0 = time.perf_counter()
    for st, fn in fn_by_stream.items():
        with torch.cuda.stream(st):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            ev[st] = (e0, e1)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    return [round(e0.elapsed_time(e1), 2) for e0, e1 in ev.values()], round(wall, 2)


def main():
    dev = torch.device("cuda")
    torch.cuda.init()
    x = torch.randn(8, 256, 64, 64, device=dev)
    w = torch.randn(256, 256, 3, 3, device=dev) * 0.02
    b = torch.zeros(256, device=dev)
    wp = ops.pack_conv3x3_weights(w)
    a = torch.randn(8, 256, 256, device=dev)
    a = a @ a.transpose(1, 2)

    def convs():
        for _ in range(40):
            ops.conv3x3(x, wp, b, 256)

    def eig():
        eigh.syevd(a)

    ALL = (1 << 256) - 1
    layouts = {
        "low16": (1 << 16) - 1,
        "2_per_32": sum(0b11 << (32 * i) for i in range(8)),
        "every16th": sum(1 << (16 * i) for i in range(16)),
        "low32": (1 << 32) - 1,
    }
    d0 = torch.cuda.current_stream()
    convs(); eig(); torch.cuda.synchronize()
    print("alone on the default stream: convs", timed({d0: convs}), " eig", timed({d0: eig}), flush=True)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s1):
        convs()
    with torch.cuda.stream(s2):
        eig()
    torch.cuda.synchronize()
    print("two plain streams together: [convs, eig], wall", timed({s1: convs, s2: eig}), flush=True)
    for name, small in layouts.items():
        big = ALL & ~small
        sa, sb = masked_stream(big), masked_stream(small)
        with torch.cuda.stream(sa):
            convs()
        with torch.cuda.stream(sb):
            eig()
        torch.cuda.synchronize()
        ca, cw = timed({sa: convs})
        ea, ew = timed({sb: eig})
        both, wall = timed({sa: convs, sb: eig})
        print(f"mask {name:10s} ({bin(small).count('1'):3d} CUs for eig): convs alone {ca[0]:7.2f}  eig alone {ea[0]:7.2f}  "
              f"together [convs, eig] {both}  wall {wall}", flush=True)


if __name__ == "__main__":
    main()
