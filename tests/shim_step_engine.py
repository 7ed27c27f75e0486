"""CPU stand-in for tla_rust_amd.sharded.HipShard, backed by tests/_shim (host build of the device lowerings + the product's level
loop, tla_rust_amd/csrc/shard_loop.h, compiled over it).  TEST ONLY: lets the multi-rank loop run under gloo without a GPU."""
import ctypes as C

import torch

import helpers
from tla_rust_amd import binding as B


class ShimShard:
    def __init__(self, spec, params, rank, world):
        self.lib = L = helpers.shim_lib()
        L.shim_shard_create.restype = C.c_void_p
        L.shim_shard_create.argtypes = [C.POINTER(helpers.McSpecDesc), C.c_uint32, C.c_uint32]
        L.shim_shard_destroy.argtypes = [C.c_void_p]
        L.shim_shard_destroy.restype = None
        L.shim_shard_run_transport.argtypes = [C.c_void_p, C.POINTER(B.Transport), C.POINTER(B.ShardOpts), C.POINTER(B.CResult)]
        L.shim_shard_trace_transport.argtypes = [C.c_void_p, C.POINTER(B.Transport), C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_size_t),
                                                 C.POINTER(C.c_int32)]
        L.shim_state_bytes.restype = C.c_size_t
        L.shim_state_bytes.argtypes = [C.POINTER(helpers.McSpecDesc)]
        L.shim_state_format.argtypes = [C.POINTER(helpers.McSpecDesc), C.c_char_p, C.c_char_p, C.c_size_t]
        L.shim_state_action_name.argtypes = [C.POINTER(helpers.McSpecDesc), C.c_char_p, C.c_int]
        L.shim_state_action_name.restype = C.c_char_p
        L.shim_state_apply.argtypes = [C.POINTER(helpers.McSpecDesc), C.c_char_p, C.c_int, C.c_char_p]
        L.shim_last_error.restype = C.c_char_p
        self.d = helpers.spec_desc(spec, params)
        self.W = L.shim_state_bytes(C.byref(self.d))
        self.h = L.shim_shard_create(C.byref(self.d), rank, world)
        self.device = torch.device("cpu")

    def run_transport(self, t, opts, res):
        return self.lib.shim_shard_run_transport(self.h, C.byref(t), C.byref(opts), C.byref(res))

    def trace_transport(self, t, states, slots, n, final):
        return self.lib.shim_shard_trace_transport(self.h, C.byref(t), states, slots, n, final)

    def checkpoint(self, path):
        self.lib.shim_shard_checkpoint.argtypes = [C.c_void_p, C.c_char_p]
        return self.lib.shim_shard_checkpoint(self.h, str(path).encode())

    def restore(self, path):
        self.lib.shim_shard_restore.argtypes = [C.c_void_p, C.c_char_p]
        return self.lib.shim_shard_restore(self.h, str(path).encode())

    def check(self, rc, what):
        if rc:
            raise RuntimeError(f"{what} failed: {rc} {self.lib.shim_last_error().decode()}")

    def format(self, st):
        buf = C.create_string_buffer(1 << 16)
        n = self.lib.shim_state_format(C.byref(self.d), st, buf, len(buf))
        return buf.raw[:n].decode()

    def action_name(self, st, slot):
        return self.lib.shim_state_action_name(C.byref(self.d), st, slot).decode()

    def apply(self, st, slot):
        out = C.create_string_buffer(self.W)
        self.lib.shim_state_apply(C.byref(self.d), st, slot, out)
        return out.raw

    def close(self):
        self.lib.shim_shard_destroy(self.h)
