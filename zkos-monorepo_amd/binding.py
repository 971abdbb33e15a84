"""ctypes binding of libshielder_hip.so (include/shielder_hip.h).

Mirrors the reference's Rust surface for the proving path:
  Params (powers_of_tau::read, commit, commit_lagrange)   /root/reference/crates/powers-of-tau/lib.rs:61-74, 247-264
  poseidon::off_circuit::hash                              /root/reference/crates/shielder_bindings/src/hash.rs:17-26
Field elements are numpy uint64 arrays (..., 4): little-endian Montgomery limbs.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "libshielder_hip.so")
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583

SHP_OK = 0
(SHP_ERR_INVALID_ARG, SHP_ERR_NO_DEVICE, SHP_ERR_HIP, SHP_ERR_FORMAT, SHP_ERR_INTERNAL, SHP_ERR_UNSATISFIED,
 SHP_ERR_BUSY, SHP_ERR_TIMEOUT) = range(1, 9)
BASIS_MONOMIAL, BASIS_LAGRANGE = 0, 1


# shp_phase_fn: (user, phase, challenges, n_challenges) -> the caller's advice block, or NULL
PHASE_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t)


class ShpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libshielder_hip error {code}: {msg}")
        self.code = code


def build(force=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_DIR, "csrc", f) for f in os.listdir(os.path.join(_DIR, "csrc"))
            if f.endswith((".hip", ".hpp", ".inc"))]
    srcs.append(os.path.join(_DIR, "..", "include", "shielder_hip.h"))
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", os.path.join(_DIR, "csrc"), "-j4"])
    return _SO


_lib = None


def set_hw_queues(n: int, quiet=False) -> int:
    """An explicit, logged choice of GPU_MAX_HW_QUEUES for this process (benchmarks, the serve front): call before
    anything touches the GPU.  Returns the value in force; a process whose HIP runtime is already up keeps its own."""
    if _lib is not None:
        return int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4)
    n = max(1, min(int(n), 32))
    before = os.environ.get("GPU_MAX_HW_QUEUES")
    os.environ["GPU_MAX_HW_QUEUES"] = str(n)
    os.environ["SHP_HW_QUEUES"] = str(n)
    if not quiet and before != str(n):
        import sys
        print(f"[shielder_hip] GPU_MAX_HW_QUEUES {before or 'unset'} -> {n} for this process", file=sys.stderr, flush=True)
    return n


def lib():
    """The HIP library; raises if it has not been built (there is no CPU fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise ShpError(-1, f"{_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "-- the product path has no CPU fallback")
        # torch (plumbing: device memory, streams) bundles its own HIP runtime and loads it RTLD_GLOBAL;
        # importing it FIRST makes libshielder_hip resolve its hip* symbols to that same runtime
        # instance, so stream handles and device pointers are interchangeable between the two.
        # Proving lanes are HIP streams; ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues (HIP's default
        # is 4), and kernels of streams that share a queue run one after the other (measured, 16 lanes: 1149
        # proofs/s on 4 queues, 1400 on 16).  The value must be in the environment before the HIP runtime
        # initialises, which torch's import below may do.  Like the library itself, this only fills in a default
        # when the variable is unset; a program that wants more says so with set_hw_queues() before its first call.
        if "SHP_HW_QUEUES" in os.environ:
            os.environ["GPU_MAX_HW_QUEUES"] = os.environ["SHP_HW_QUEUES"]
        elif not os.environ.get("GPU_MAX_HW_QUEUES"):
            os.environ["GPU_MAX_HW_QUEUES"] = "16"
        import torch  # noqa: F401
        _lib = C.CDLL(_SO)
        _lib.shp_last_error.restype = C.c_char_p
        _lib.shp_version.restype = C.c_char_p
    return _lib


def _check(rc):
    if rc != SHP_OK:
        raise ShpError(rc, lib().shp_last_error().decode())


def _vp(x):
    return C.c_void_p(int(x))


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---------------------------------------------------------------- host field helpers (no GPU)
def fr_from_le32(b: bytes):
    out = (C.c_uint64 * 4)()
    _check(lib().shp_fr_from_le32(C.c_char_p(bytes(b)), out))
    return np.array(list(out), dtype=np.uint64)


def fr_to_le32(limbs) -> bytes:
    a = (C.c_uint64 * 4)(*[int(v) for v in np.asarray(limbs, dtype=np.uint64).reshape(4)])
    out = C.create_string_buffer(32)
    _check(lib().shp_fr_to_le32(a, out))
    return out.raw


def to_device(arr):
    """numpy uint64 (...,4|8) -> torch int64 CUDA tensor with the same bytes"""
    import torch
    a = np.ascontiguousarray(arr, dtype=np.uint64)
    return torch.from_numpy(a.view(np.int64)).cuda()


def to_host(t):
    return t.detach().cpu().numpy().view(np.uint64)


def device_count() -> int:
    n = C.c_int32()
    _check(lib().shp_device_count(C.byref(n)))
    return n.value


class Context:
    """One per (process, GPU): shp_ctx."""

    created = 0          # contexts made by this process: once > 0 the process has initialised the GPU

    def __init__(self, device=0):
        self.h = C.c_void_p()
        _check(lib().shp_ctx_create(C.c_int32(device), C.byref(self.h)))
        Context.created += 1
        self.device = device
        # torch is plumbing only: make the context's HIP stream torch's current stream, so that
        # torch copies, torch.cuda.Event timing and the library's kernels are ordered on ONE stream
        import torch
        sp = C.c_void_p()
        _check(lib().shp_ctx_stream(self.h, C.byref(sp)))
        torch.cuda.set_device(device)
        self._prev_stream = torch.cuda.current_stream(device)
        self.torch_stream = torch.cuda.ExternalStream(sp.value, device=device)
        torch.cuda.set_stream(self.torch_stream)

    def close(self):
        if self.h:
            # torch must not be left with a current stream that no longer exists (a second context of a process -- one per
            # GPU in the serve front -- comes and goes while the first one lives on)
            import torch
            if torch.cuda.current_stream(self.device) == self.torch_stream:
                torch.cuda.set_stream(self._prev_stream)
            lib().shp_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def sync(self):
        _check(lib().shp_ctx_sync(self.h))

    def set_hbm_limit(self, nbytes: int):
        """cap on the device memory the process may hold while MSM tables are sized (0: none)"""
        _check(lib().shp_ctx_set_hbm_limit(self.h, C.c_uint64(int(nbytes))))

    def fr_mul_rate(self, iters=4096):
        out = C.c_double()
        _check(lib().shp_diag_fr_mul_rate(self.h, C.c_uint32(iters), C.byref(out)))
        return out.value

    # -- NTT -------------------------------------------------------------------------
    def ntt(self, t, log_n, batch=1, inverse=False):
        """in place on a torch int64 CUDA tensor holding batch x 2^log_n Fr"""
        assert t.is_cuda and t.numel() == batch * (1 << log_n) * 4
        _check(lib().shp_ntt_fr(self.h, _vp(t.data_ptr()), C.c_uint32(log_n), C.c_uint32(batch),
                                C.c_int32(1 if inverse else 0), _stream()))
        return t

    def ntt_noalloc(self, t, log_n, batch=1, inverse=False):
        """shp_ntt_fr_noalloc: the same transform through the library's static ping-pong array (no allocation)"""
        assert t.is_cuda and t.numel() == batch * (1 << log_n) * 4
        _check(lib().shp_ntt_fr_noalloc(self.h, _vp(t.data_ptr()), C.c_uint32(log_n), C.c_uint32(batch),
                                        C.c_int32(1 if inverse else 0), _stream()))
        return t

    def coeff_to_extended(self, t_in, k, ext_k, batch=1):
        import torch
        out = torch.empty(batch * (1 << ext_k) * 4, dtype=torch.int64, device=t_in.device)
        _check(lib().shp_coeff_to_extended(self.h, _vp(t_in.data_ptr()), _vp(out.data_ptr()), C.c_uint32(k),
                                           C.c_uint32(ext_k), C.c_uint32(batch), _stream()))
        return out

    def extended_to_coeff(self, t, ext_k, batch=1):
        _check(lib().shp_extended_to_coeff(self.h, _vp(t.data_ptr()), C.c_uint32(ext_k), C.c_uint32(batch), _stream()))
        return t

    # -- Poseidon2 ---------------------------------------------------------------------
    def poseidon2_hash(self, t_in, n_hashes, arity):
        import torch
        out = torch.empty(n_hashes * 4, dtype=torch.int64, device=t_in.device)
        _check(lib().shp_poseidon2_t8(self.h, _vp(t_in.data_ptr()), C.c_size_t(n_hashes), C.c_uint32(arity),
                                      _vp(out.data_ptr()), _stream()))
        return out

    def poseidon2_permute(self, t_states, n_states):
        _check(lib().shp_poseidon2_t8_permute(self.h, _vp(t_states.data_ptr()), C.c_size_t(n_states), _stream()))
        return t_states


class Msm:
    """Fixed-base MSM plan (shp_msm)."""

    def __init__(self, ctx: Context, bases, window_bits=0):
        bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 8)
        self.ctx, self.n = ctx, bases.shape[0]
        self.h = C.c_void_p()
        _check(lib().shp_msm_create(ctx.h, bases.ctypes.data_as(C.c_void_p), C.c_size_t(self.n),
                                    C.c_uint32(window_bits), C.byref(self.h)))

    @classmethod
    def from_device(cls, ctx: Context, t_bases, n, window_bits=0):
        self = cls.__new__(cls)
        self.ctx, self.n = ctx, n
        self.h = C.c_void_p()
        _check(lib().shp_msm_create_dev(ctx.h, _vp(t_bases.data_ptr()), C.c_size_t(n), C.c_uint32(window_bits),
                                        C.byref(self.h)))
        return self

    def run(self, t_scalars, n_scalars, batch=1, stride=0):
        out = np.zeros((batch, 8), dtype=np.uint64)
        _check(lib().shp_msm_g1(self.h, _vp(t_scalars.data_ptr()), C.c_size_t(n_scalars), C.c_size_t(stride),
                                C.c_uint32(batch), out.ctypes.data_as(C.c_void_p), _stream()))
        return out

    def stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        _check(lib().shp_msm_last_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        if self.h:
            lib().shp_msm_destroy(self.h)
            self.h = C.c_void_p()


def distinct_points(ctx: Context, bases, multiples: int):
    """shp_diag_multiples: (rows, 8) uint64 affine points -> CUDA tensor of rows * multiples DISTINCT affine points
    d * bases[r], d = 1 .. multiples (synthetic inputs of the stand-alone MSM measurements, generated on the device)"""
    import torch
    b = to_device(np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 8))
    rows = b.shape[0]
    out = torch.empty((rows * multiples, 8), dtype=torch.int64, device=b.device)
    _check(lib().shp_diag_multiples(ctx.h, _vp(b.data_ptr()), C.c_size_t(rows), C.c_uint32(multiples), _vp(out.data_ptr())))
    return out


class Params:
    """Device-resident KZG SRS: the reference's `Params` (ParamsKZG<Bn256>)."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle
        k = C.c_uint32()
        _check(lib().shp_srs_k(self.h, C.byref(k)))
        self.k, self.n = k.value, 1 << k.value

    @classmethod
    def read_ptau(cls, ctx, data: bytes):
        h = C.c_void_p()
        _check(lib().shp_srs_from_ptau(ctx.h, C.c_char_p(data), C.c_size_t(len(data)), C.byref(h)))
        return cls(ctx, h)

    def msm_info(self, basis=1):
        """(direct, window_bits, table_bytes) of the SRS's MSM plan for one basis"""
        d, c, b = C.c_int32(), C.c_uint32(), C.c_uint64()
        _check(lib().shp_srs_msm_info(self.h, C.c_int32(basis), C.byref(d), C.byref(c), C.byref(b)))
        return bool(d.value), c.value, b.value

    def downsize(self, k: int):
        """Params::downsize: a new Params over the first 2^k powers (reference shielder-cli pk.rs:110)"""
        h = C.c_void_p()
        _check(lib().shp_srs_downsize(self.h, C.c_uint32(k), C.byref(h)))
        return type(self)(self.ctx, h)

    @classmethod
    def read_raw(cls, ctx, data: bytes):
        h = C.c_void_p()
        _check(lib().shp_srs_from_halo2_raw(ctx.h, C.c_char_p(data), C.c_size_t(len(data)), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_parts(cls, ctx, k, g, g_lagrange=None, g2=None, s_g2=None):
        g = np.ascontiguousarray(g, dtype=np.uint64)
        gl = None if g_lagrange is None else np.ascontiguousarray(g_lagrange, dtype=np.uint64)
        h = C.c_void_p()
        _check(lib().shp_srs_from_parts(ctx.h, C.c_uint32(k), g.ctypes.data_as(C.c_void_p),
                                        gl.ctypes.data_as(C.c_void_p) if gl is not None else None,
                                        C.c_char_p(g2) if g2 else None, C.c_char_p(s_g2) if s_g2 else None,
                                        C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def synthetic(cls, ctx, k, tau: int):
        """generate_setup_params stand-in: SRS with a known trapdoor (tests / benchmarks)"""
        h = C.c_void_p()
        _check(lib().shp_srs_synthetic(ctx.h, C.c_uint32(k), C.c_char_p(int(tau).to_bytes(32, "little")), C.byref(h)))
        return cls(ctx, h)

    def get_g(self, basis=BASIS_MONOMIAL):
        out = np.zeros((self.n, 8), dtype=np.uint64)
        _check(lib().shp_srs_get_g(self.h, C.c_int32(basis), out.ctypes.data_as(C.c_void_p), C.c_size_t(self.n)))
        return out

    def save_raw(self) -> bytes:
        """marshall_params: halo2 RawBytes (k | g | g_lagrange | g2 | s_g2), what read_raw reads"""
        n = C.c_size_t()
        _check(lib().shp_srs_save(self.h, None, C.c_size_t(0), C.byref(n)))
        out = C.create_string_buffer(n.value)
        _check(lib().shp_srs_save(self.h, out, C.c_size_t(n.value), C.byref(n)))
        return out.raw[:n.value]

    def g2(self):
        a, b = C.create_string_buffer(128), C.create_string_buffer(128)
        _check(lib().shp_srs_get_g2(self.h, a, b))
        return a.raw, b.raw

    def commit(self, t_polys, batch=1, basis=BASIS_MONOMIAL):
        out = np.zeros((batch, 8), dtype=np.uint64)
        _check(lib().shp_srs_commit(self.h, C.c_int32(basis), _vp(t_polys.data_ptr()), C.c_uint32(batch),
                                    out.ctypes.data_as(C.c_void_p), _stream()))
        return out

    def commit_lagrange(self, t_polys, batch=1):
        return self.commit(t_polys, batch, BASIS_LAGRANGE)

    def close(self):
        if self.h:
            lib().shp_srs_destroy(self.h)
            self.h = C.c_void_p()


def poseidon2_hash_host(ints):
    """poseidon::off_circuit::hash on canonical python ints (host path of the library)"""
    buf = b"".join(int(x).to_bytes(32, "little") for x in ints)
    out = C.create_string_buffer(32)
    _check(lib().shp_poseidon2_hash_host(C.c_char_p(buf), C.c_uint32(len(ints)), out))
    return int.from_bytes(out.raw, "little")


def blinding_stream(seed: bytes, first: int, count: int):
    """the Fr draws shp_prove(seed) consumes: draw i = from_u512(ChaCha20(seed, block i)); (count, 4) Montgomery limbs"""
    out = np.zeros((count, 4), dtype=np.uint64)
    _check(lib().shp_blinding_stream(C.c_char_p(seed), C.c_uint64(first), C.c_size_t(count), out.ctypes.data_as(C.c_void_p)))
    return out


def plan_min_k(plan: bytes) -> int:
    k = C.c_uint32()
    _check(lib().shp_plan_min_k(C.c_char_p(plan), C.c_size_t(len(plan)), C.byref(k)))
    return k.value


def plan_compile(description: bytes) -> bytes:
    """shp_plan_compile: circuit description (PlanBuilder.build_description) -> circuit plan"""
    n = C.c_size_t()
    _check(lib().shp_plan_compile(C.c_char_p(description), C.c_size_t(len(description)), None, C.c_size_t(0), C.byref(n)))
    out = C.create_string_buffer(n.value)
    _check(lib().shp_plan_compile(C.c_char_p(description), C.c_size_t(len(description)), out, C.c_size_t(n.value), C.byref(n)))
    return out.raw[:n.value]


def plan_retarget(plan: bytes, k: int) -> bytes:
    n = C.c_size_t()
    _check(lib().shp_plan_retarget(C.c_char_p(plan), C.c_size_t(len(plan)), C.c_uint32(k), None, C.c_size_t(0), C.byref(n)))
    out = C.create_string_buffer(n.value)
    _check(lib().shp_plan_retarget(C.c_char_p(plan), C.c_size_t(len(plan)), C.c_uint32(k), out, C.c_size_t(n.value), C.byref(n)))
    return out.raw[:n.value]


class ProvingKey:
    """Device-resident proving key of one circuit plan (the reference's `ProvingKey` + vk)."""

    def __init__(self, ctx: Context, params: Params, plan: bytes = None, saved: bytes = None, transcript_repr: int = None,
                 handle=None):
        """keygen (plan given; transcript_repr = halo2's vk.transcript_repr() when the caller has it) or unmarshall_pk
        (saved = bytes of save())"""
        self.ctx, self.params = ctx, params
        self.h = C.c_void_p()
        if handle is not None:
            self.h = handle
        elif saved is not None:
            _check(lib().shp_pk_load(ctx.h, params.h, C.c_char_p(saved), C.c_size_t(len(saved)), C.byref(self.h)))
        elif transcript_repr is not None:
            _check(lib().shp_keygen_with_transcript_repr(ctx.h, params.h, C.c_char_p(plan), C.c_size_t(len(plan)),
                                                         C.c_char_p(int(transcript_repr).to_bytes(32, "little")), C.byref(self.h)))
        else:
            _check(lib().shp_keygen(ctx.h, params.h, C.c_char_p(plan), C.c_size_t(len(plan)), C.byref(self.h)))
        k, ni, nins, pl = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(lib().shp_pk_info(self.h, C.byref(k), C.byref(ni), C.byref(nins), C.byref(pl)))
        self.k, self.n_inputs, self.n_instances, self.proof_len = k.value, ni.value, nins.value, pl.value
        nd = C.c_size_t()
        _check(lib().shp_pk_randomness_len(self.h, C.byref(nd)))
        self.n_draws = nd.value

    @classmethod
    def load(cls, ctx, params, saved: bytes):
        return cls(ctx, params, saved=saved)

    @classmethod
    def generate_keys_with_min_k(cls, ctx, full_params: Params, plan: bytes, transcript_repr: int = None):
        """generate_keys_with_min_k(circuit, full_params) -> (params, k, pk)
        (call site /root/reference/crates/shielder_bindings/build.rs:22-31), in the library: the smallest k that holds the
        circuit, the plan laid out for it, the SRS downsized (Params::downsize), the keys generated over the result."""
        hs, hp, k = C.c_void_p(), C.c_void_p(), C.c_uint32()
        tr = C.c_char_p(int(transcript_repr).to_bytes(32, "little")) if transcript_repr is not None else None
        _check(lib().shp_generate_keys_with_min_k(ctx.h, full_params.h, C.c_char_p(plan), C.c_size_t(len(plan)), tr,
                                                  C.byref(hs), C.byref(k), C.byref(hp)))
        params = Params(ctx, hs)
        return params, k.value, cls(ctx, params, handle=hp)

    def plan(self) -> bytes:
        """the circuit plan the key was generated from (re-laid-out by generate_keys_with_min_k)"""
        n = C.c_size_t()
        _check(lib().shp_pk_plan(self.h, None, C.c_size_t(0), C.byref(n)))
        out = C.create_string_buffer(n.value)
        _check(lib().shp_pk_plan(self.h, out, C.c_size_t(n.value), C.byref(n)))
        return out.raw[:n.value]

    def witness_columns(self, witness_ints):
        """shp_witness_columns: the plan's witness program alone, on the host -> (advice (A, n, 4) uint64 Montgomery limbs,
        instances (n_instances, 4), leading rows the program assigns): what shp_prove_assigned takes"""
        wit = b"".join(int(x).to_bytes(32, "little") for x in witness_ints)
        A, n = len(self.phases()[2]), 1 << self.k
        adv = np.zeros((A, n, 4), dtype=np.uint64)
        ins = np.zeros((max(self.n_instances, 1), 4), dtype=np.uint64)
        rows = C.c_size_t()
        _check(lib().shp_witness_columns(self.h, C.c_char_p(wit), C.c_size_t(len(witness_ints)), adv.ctypes.data_as(C.c_void_p),
                                         C.c_size_t(A), C.c_size_t(n), ins.ctypes.data_as(C.c_void_p), C.c_size_t(self.n_instances),
                                         C.byref(rows)))
        return adv, ins[:self.n_instances], rows.value

    def save(self) -> bytes:
        """marshall_pk: plan + verifying-key material (the rest is rebuilt on the GPU at load time)"""
        n = C.c_size_t()
        _check(lib().shp_pk_save(self.h, None, C.c_size_t(0), C.byref(n)))
        out = C.create_string_buffer(n.value)
        _check(lib().shp_pk_save(self.h, out, C.c_size_t(n.value), C.byref(n)))
        return out.raw[:n.value]

    def public_inputs(self, witness_ints):
        """*_pub_inputs of the bindings: the circuit's public inputs for a witness, without proving"""
        wit = b"".join(int(x).to_bytes(32, "little") for x in witness_ints)
        out = C.create_string_buffer(max(1, 32 * self.n_instances))
        _check(lib().shp_public_inputs(self.h, C.c_char_p(wit), C.c_size_t(len(witness_ints)), out, C.c_size_t(len(out.raw))))
        return [int.from_bytes(out.raw[32 * i:32 * i + 32], "little") for i in range(self.n_instances)]

    def verify(self, proof: bytes, instances) -> bool:
        """Circuit::verify (crates/shielder_bindings/src/circuits/mod.rs:113-124): native verifier incl. pairing"""
        ins = b"".join(int(x).to_bytes(32, "little") for x in instances)
        ok = C.c_int32(0)
        _check(lib().shp_verify(self.h, C.c_char_p(proof), C.c_size_t(len(proof)), C.c_char_p(ins), C.c_size_t(len(instances)),
                                C.byref(ok)))
        return bool(ok.value)

    def verify_trace(self, proof: bytes, instances, with_challenges=False):
        """shp_verify_trace: (accepted, [22 ints]) -- the words the reference's contract keeps (Halo2Verifier.sol:39-73);
        with_challenges: (accepted, [22 ints], the circuit's own challenges) for a circuit with several phases"""
        ins = b"".join(int(x).to_bytes(32, "little") for x in instances)
        ok = C.c_int32(0)
        nw = 22 + (self.phases()[1] if with_challenges else 0)
        words = C.create_string_buffer(nw * 32)
        _check(lib().shp_verify_trace(self.h, C.c_char_p(proof), C.c_size_t(len(proof)), C.c_char_p(ins), C.c_size_t(len(instances)),
                                      C.byref(ok), words, C.c_size_t(nw * 32)))
        w = [int.from_bytes(words.raw[32 * i:32 * i + 32], "big") for i in range(nw)]
        return (bool(ok.value), w[:22], w[22:]) if with_challenges else (bool(ok.value), w)

    def phases(self):
        """shp_pk_phases: (number of phases, number of challenges, phase of every advice column)"""
        nph, nch, na = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(lib().shp_pk_phases(self.h, C.byref(nph), C.byref(nch), C.byref(na), None, C.c_size_t(0)))
        cols = (C.c_uint32 * max(na.value, 1))()
        _check(lib().shp_pk_phases(self.h, None, None, None, cols, C.c_size_t(na.value)))
        return nph.value, nch.value, list(cols)[:na.value]

    def gate_compiled(self) -> bool:
        """True when the key's gate expressions run as a kernel compiled ahead of time (a plan the library ships)"""
        v = C.c_int32(0)
        _check(lib().shp_pk_gate_compiled(self.h, C.byref(v)))
        return bool(v.value)

    def compiled_level(self) -> int:
        """0: the quotient's expressions are interpreted; 1: the gate program runs as a compiled kernel; 2: the whole
        numerator of the quotient (gates, permutation and lookup terms) runs as one compiled kernel"""
        v = C.c_int32(0)
        _check(lib().shp_pk_gate_compiled(self.h, C.byref(v)))
        return v.value

    def witness_native(self):
        """(ranges, ops, total_ops): the ranges of the witness program that run as native Poseidon2 permutations, the
        operations they cover, the program's length"""
        r, o, t = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        _check(lib().shp_pk_witness_native(self.h, C.byref(r), C.byref(o), C.byref(t)))
        return r.value, o.value, t.value

    def vk_digest(self) -> bytes:
        out = C.create_string_buffer(32)
        _check(lib().shp_pk_vk_digest(self.h, out))
        return out.raw

    def commitments(self, n_fixed, n_sigma):
        f = np.zeros((n_fixed, 8), dtype=np.uint64)
        s = np.zeros((n_sigma, 8), dtype=np.uint64)
        _check(lib().shp_pk_commitments(self.h, f.ctypes.data_as(C.c_void_p), C.c_size_t(n_fixed),
                                        s.ctypes.data_as(C.c_void_p), C.c_size_t(n_sigma)))
        return f, s

    def close(self):
        if self.h:
            lib().shp_pk_destroy(self.h)
            self.h = C.c_void_p()


class Prover:
    """One proving lane (HIP stream + HBM workspace): generate_proof(params, pk, circuit, pub_input, rng)."""

    def __init__(self, pk: ProvingKey):
        self.pk = pk
        self.h = C.c_void_p()
        _check(lib().shp_prover_create(pk.h, C.byref(self.h)))

    def prove(self, witness_ints, seed: bytes):
        """witness_ints: canonical python ints in the plan's witness-input order -> (proof, instances)"""
        assert len(seed) == 32
        return self.prove_bytes(b"".join(int(x).to_bytes(32, "little") for x in witness_ints), seed)

    def prove_bytes(self, wit: bytes, seed: bytes):
        proof = C.create_string_buffer(self.pk.proof_len)
        plen = C.c_size_t()
        inst = C.create_string_buffer(32 * self.pk.n_instances)
        _check(lib().shp_prove(self.h, C.c_char_p(wit), C.c_size_t(len(wit) // 32), C.c_char_p(seed), proof,
                               C.c_size_t(self.pk.proof_len), C.byref(plen), inst, C.c_size_t(len(inst.raw))))
        ins = [int.from_bytes(inst.raw[32 * i:32 * i + 32], "little") for i in range(self.pk.n_instances)]
        return proof.raw[:plen.value], ins

    def prove_with_randomness(self, witness_ints, draws):
        """generate_proof with the caller's pre-drawn Fr::random values: draws = (pk.n_draws, 4) uint64 Montgomery limbs"""
        wit = b"".join(int(x).to_bytes(32, "little") for x in witness_ints)
        d = np.ascontiguousarray(draws, dtype=np.uint64)
        proof = C.create_string_buffer(self.pk.proof_len)
        plen = C.c_size_t()
        inst = C.create_string_buffer(max(1, 32 * self.pk.n_instances))
        _check(lib().shp_prove_with_randomness(self.h, C.c_char_p(wit), C.c_size_t(len(wit) // 32), d.ctypes.data_as(C.c_void_p),
                                               C.c_size_t(d.shape[0]), proof, C.c_size_t(self.pk.proof_len), C.byref(plen), inst,
                                               C.c_size_t(len(inst.raw))))
        ins = [int.from_bytes(inst.raw[32 * i:32 * i + 32], "little") for i in range(self.pk.n_instances)]
        return proof.raw[:plen.value], ins

    def prove_assigned(self, advice, instances, assigned_rows=0, seed: bytes = None, draws=None):
        """generate_proof for caller-synthesised advice columns: advice = (A, n, 4) uint64 Montgomery limbs, instances =
        (n_instances, 4); randomness = `draws` ((pk.n_draws, 4) Montgomery) or a 32-byte `seed`"""
        adv = np.ascontiguousarray(advice, dtype=np.uint64)
        ins = np.ascontiguousarray(instances, dtype=np.uint64).reshape(-1, 4)
        d = None if draws is None else np.ascontiguousarray(draws, dtype=np.uint64)
        proof = C.create_string_buffer(self.pk.proof_len)
        plen = C.c_size_t()
        _check(lib().shp_prove_assigned(self.h, adv.ctypes.data_as(C.c_void_p), C.c_size_t(adv.shape[0]), C.c_size_t(adv.shape[1]),
                                        C.c_size_t(assigned_rows), ins.ctypes.data_as(C.c_void_p), C.c_size_t(ins.shape[0]),
                                        d.ctypes.data_as(C.c_void_p) if d is not None else None,
                                        C.c_size_t(d.shape[0] if d is not None else 0),
                                        C.c_char_p(seed) if seed is not None else None, proof, C.c_size_t(self.pk.proof_len),
                                        C.byref(plen)))
        return proof.raw[:plen.value]

    def prove_assigned_phased(self, synthesize, instances, assigned_rows=0, seed: bytes = None, draws=None):
        """generate_proof for a circuit with several phases whose columns the caller synthesises: `synthesize(phase,
        challenges)` is called once per phase with the challenges squeezed so far (ints; those of later phases are 0) and
        returns the advice block ((A, n, 4) uint64 Montgomery limbs) of which the library reads that phase's columns"""
        ins = np.ascontiguousarray(instances, dtype=np.uint64).reshape(-1, 4)
        d = None if draws is None else np.ascontiguousarray(draws, dtype=np.uint64)
        keep, failure = [], []

        @PHASE_FN
        def trampoline(_user, phase, chal_ptr, n_chal):
            try:
                limbs = np.ctypeslib.as_array(C.cast(chal_ptr, C.POINTER(C.c_uint64)), shape=(n_chal, 4)).copy() if n_chal else np.zeros((0, 4), np.uint64)
                chal = [int.from_bytes(fr_to_le32(limbs[i]), "little") for i in range(n_chal)]
                block = np.ascontiguousarray(synthesize(int(phase), chal), dtype=np.uint64)
                keep[:] = [block]                      # valid until the next call
                return block.ctypes.data
            except BaseException as e:                 # an exception must not cross the C frames: the proof is abandoned
                failure.append(e)
                return None
        ncols, nrows = len(self.pk.phases()[2]), 1 << self.pk.k
        proof = C.create_string_buffer(self.pk.proof_len)
        plen = C.c_size_t()
        rc = lib().shp_prove_assigned_phased(self.h, trampoline, None, C.c_size_t(ncols), C.c_size_t(nrows), C.c_size_t(assigned_rows),
                                             ins.ctypes.data_as(C.c_void_p), C.c_size_t(ins.shape[0]),
                                             d.ctypes.data_as(C.c_void_p) if d is not None else None,
                                             C.c_size_t(d.shape[0] if d is not None else 0),
                                             C.c_char_p(seed) if seed is not None else None, proof, C.c_size_t(self.pk.proof_len),
                                             C.byref(plen))
        if failure:
            raise failure[0]
        _check(rc)
        return proof.raw[:plen.value]

    def mode(self, throughput: bool):
        """shp_prover_mode: latency (default) or throughput scheduling of this lane"""
        _check(lib().shp_prover_mode(self.h, C.c_uint32(1 if throughput else 0)))

    def witness_threads(self, n: int):
        """host threads for the witness program of this lane (default 4; 1 = sequential)"""
        _check(lib().shp_prover_witness_threads(self.h, C.c_uint32(n)))

    def profile(self, enable=True):
        """True / 1: HIP-event kernel timers + phase clock; 2: the phase clock alone; False / 0: off"""
        _check(lib().shp_prover_profile(self.h, C.c_int32(int(enable))))

    def phases(self):
        """host phase clock of the last proof (profile(True) first): [(name, microseconds)] in proof order"""
        out, i = [], 0
        while True:
            name, us = C.c_char_p(), C.c_double()
            if lib().shp_prover_phase_read(self.h, C.c_uint32(i), C.byref(name), C.byref(us)) != SHP_OK:
                return out
            out.append((name.value.decode(), us.value))
            i += 1

    def profile_read(self):
        """{kernel family: (launches, total_ms)} measured with HIP events on the lane's stream"""
        out = {}
        for i in range(64):
            name, cnt, ms = C.c_char_p(), C.c_uint64(), C.c_double()
            if lib().shp_prover_profile_read(self.h, C.c_uint32(i), C.byref(name), C.byref(cnt), C.byref(ms)) != SHP_OK:
                break
            out[name.value.decode()] = (cnt.value, ms.value)
        return out

    def close(self):
        if self.h:
            lib().shp_prover_destroy(self.h)
            self.h = C.c_void_p()


class MerkleTree:
    """Arity-7 Poseidon2 Merkle tree on the GPU (contracts/MerkleTree.sol): build over leaves resident in
    HBM, then getMerklePath for batches of leaves."""

    def __init__(self, ctx: Context, t_leaves, n_leaves: int, height: int):
        self.height = height
        self.h = C.c_void_p()
        _check(lib().shp_merkle_build(ctx.h, _vp(t_leaves.data_ptr()), C.c_size_t(n_leaves), C.c_uint32(height), C.byref(self.h)))

    def root(self) -> int:
        out = C.create_string_buffer(32)
        _check(lib().shp_merkle_root(self.h, out))
        return int.from_bytes(out.raw, "little")

    def paths(self, indices):
        """-> list of (path[height][7] ints, root) in the reference's reorganize_merkle_path shape"""
        idx = (C.c_uint64 * len(indices))(*indices)
        per = 7 * self.height + 1
        out = C.create_string_buffer(32 * per * len(indices))
        _check(lib().shp_merkle_paths(self.h, idx, C.c_size_t(len(indices)), out))
        res = []
        for q in range(len(indices)):
            flat = [int.from_bytes(out.raw[32 * (q * per + e): 32 * (q * per + e + 1)], "little") for e in range(per)]
            res.append(([flat[7 * l: 7 * l + 7] for l in range(self.height)], flat[-1]))
        return res

    def close(self):
        if self.h:
            lib().shp_merkle_destroy(self.h)
            self.h = C.c_void_p()


def derive_nullifier(ident: int, nonce: int) -> int:
    """shielder_account::secrets::nonced::derive_nullifier (secrets.rs:37-44)"""
    out = C.create_string_buffer(32)
    _check(lib().shp_derive_nullifier(C.c_char_p(int(ident).to_bytes(32, "little")), C.c_uint32(nonce), out))
    return int.from_bytes(out.raw, "little")


def derive_id(private_key: int, chain_id: int, account_nonce: int) -> int:
    """shielder_account::secrets::derive_id (secrets.rs:48-55)"""
    out = C.create_string_buffer(32)
    _check(lib().shp_derive_id(C.c_char_p(int(private_key).to_bytes(32, "little")), C.c_uint64(chain_id), C.c_uint32(account_nonce), out))
    return int.from_bytes(out.raw, "little")


def encode_calldata(proof: bytes, instances) -> bytes:
    """halo2_verifier::encode_calldata: calldata of Halo2Verifier.verifyProof(bytes,uint256[])
    (/root/reference/crates/halo2-verifier/src/lib/verifier_contract.rs:13-20)"""
    ins = b"".join(int(x).to_bytes(32, "little") for x in instances)
    n = C.c_size_t()
    _check(lib().shp_encode_calldata(C.c_char_p(proof), C.c_size_t(len(proof)), C.c_char_p(ins), C.c_size_t(len(instances)),
                                     None, C.c_size_t(0), C.byref(n)))
    out = C.create_string_buffer(n.value)
    _check(lib().shp_encode_calldata(C.c_char_p(proof), C.c_size_t(len(proof)), C.c_char_p(ins), C.c_size_t(len(instances)),
                                     out, C.c_size_t(n.value), C.byref(n)))
    return out.raw[:n.value]


class Pool:
    """N proving lanes of one device-resident key behind a bounded FIFO queue (shp_pool): the native
    serve path -- replaces the TEE's per-connection proving tasks
    (/root/reference/tee/crates/shielder-prover-tee/src/main.rs:13-20, server.rs:157-195)."""

    def __init__(self, pk: ProvingKey, lanes: int, queue_cap: int = 0):
        self.pk = pk
        self.lanes = lanes
        self.h = C.c_void_p()
        _check(lib().shp_pool_create(pk.h, C.c_uint32(lanes), C.c_uint32(queue_cap), C.byref(self.h)))

    def lane(self, i):
        """borrowed Prover view of lane i (profiling); do not close it"""
        pr = Prover.__new__(Prover)
        pr.pk = self.pk
        pr.h = C.c_void_p()
        _check(lib().shp_pool_lane(self.h, C.c_uint32(i), C.byref(pr.h)))
        pr.close = lambda: None
        return pr

    def submit_assigned(self, advice, instances, assigned_rows=0, seed: bytes = None, draws=None) -> int:
        """shp_pool_submit_assigned: a job from caller-synthesised columns (the arguments of Prover.prove_assigned)"""
        adv = np.ascontiguousarray(advice, dtype=np.uint64)
        ins = np.ascontiguousarray(instances, dtype=np.uint64).reshape(-1, 4)
        d = None if draws is None else np.ascontiguousarray(draws, dtype=np.uint64)
        t = C.c_uint64()
        _check(lib().shp_pool_submit_assigned(self.h, adv.ctypes.data_as(C.c_void_p), C.c_size_t(adv.shape[0]), C.c_size_t(adv.shape[1]),
                                              C.c_size_t(assigned_rows), ins.ctypes.data_as(C.c_void_p), C.c_size_t(ins.shape[0]),
                                              d.ctypes.data_as(C.c_void_p) if d is not None else None,
                                              C.c_size_t(d.shape[0] if d is not None else 0),
                                              C.c_char_p(seed) if seed is not None else None, C.byref(t)))
        return t.value

    def submit(self, wit: bytes, seed: bytes) -> int:
        t = C.c_uint64()
        _check(lib().shp_pool_submit(self.h, C.c_char_p(wit), C.c_size_t(len(wit) // 32), C.c_char_p(seed), C.byref(t)))
        return t.value

    def wait(self, ticket: int):
        proof = C.create_string_buffer(self.pk.proof_len)
        plen = C.c_size_t()
        inst = C.create_string_buffer(32 * self.pk.n_instances)
        _check(lib().shp_pool_wait(self.h, C.c_uint64(ticket), proof, C.c_size_t(self.pk.proof_len), C.byref(plen), inst,
                                   C.c_size_t(len(inst.raw))))
        return proof.raw[:plen.value], [int.from_bytes(inst.raw[32 * i:32 * i + 32], "little") for i in range(self.pk.n_instances)]

    def wait_for(self, ticket: int, timeout_ms: int):
        """wait bounded by the reference's run timeout: ShpError(SHP_ERR_TIMEOUT) abandons the ticket"""
        proof = C.create_string_buffer(self.pk.proof_len)
        plen = C.c_size_t()
        inst = C.create_string_buffer(32 * self.pk.n_instances)
        _check(lib().shp_pool_wait_for(self.h, C.c_uint64(ticket), C.c_uint32(timeout_ms), proof, C.c_size_t(self.pk.proof_len),
                                       C.byref(plen), inst, C.c_size_t(len(inst.raw))))
        return proof.raw[:plen.value], [int.from_bytes(inst.raw[32 * i:32 * i + 32], "little") for i in range(self.pk.n_instances)]

    def outstanding(self) -> int:
        """jobs accepted and not finished (queued or on a lane)"""
        n = C.c_uint64()
        _check(lib().shp_pool_outstanding(self.h, C.byref(n)))
        return n.value

    def prove_batch(self, wits: bytes, seeds: bytes, count: int):
        """count proofs: wits = count witnesses back to back, seeds = count x 32 bytes -> [(proof, instances)]"""
        n_in = len(wits) // 32 // count
        stride, istride = self.pk.proof_len, 32 * self.pk.n_instances
        proofs = C.create_string_buffer(stride * count)
        insts = C.create_string_buffer(max(1, istride * count))
        lens = (C.c_size_t * count)()
        _check(lib().shp_prove_batch(self.h, C.c_char_p(wits), C.c_size_t(n_in), C.c_size_t(count), C.c_char_p(seeds), proofs,
                                     C.c_size_t(stride), lens, insts, C.c_size_t(istride)))
        out = []
        for i in range(count):
            ins = [int.from_bytes(insts.raw[istride * i + 32 * j: istride * i + 32 * j + 32], "little") for j in range(self.pk.n_instances)]
            out.append((proofs.raw[stride * i: stride * i + lens[i]], ins))
        return out

    def prove_batch_raw(self, wits: bytes, seeds: bytes, count: int, proofs_buf, lens_buf):
        """same without unpacking (bench): buffers are caller-owned ctypes arrays"""
        n_in = len(wits) // 32 // count
        _check(lib().shp_prove_batch(self.h, C.c_char_p(wits), C.c_size_t(n_in), C.c_size_t(count), C.c_char_p(seeds), proofs_buf,
                                     C.c_size_t(self.pk.proof_len), lens_buf, None, C.c_size_t(0)))

    def stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        _check(lib().shp_pool_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        if self.h:
            lib().shp_pool_destroy(self.h)
            self.h = C.c_void_p()


# ---- TEE wire protocol, native front door (csrc/wire.hip) ---------------------------------------------
TEE_PING, TEE_PUBLIC_KEY, TEE_GENERATE_PROOF = 0, 1, 2
TEE_CIRCUITS = ("NewAccount", "Deposit", "Withdraw")                     # CircuitType, protocol.rs:62-68


def tee_decode_request(body: bytes, height: int):
    """one frame body -> (kind, circuit name or None, witness bytes or None); ShpError(SHP_ERR_FORMAT) on a bad frame"""
    kind, circuit, n = C.c_int32(), C.c_int32(), C.c_size_t()
    cap = 16 * 32 + 32 * 7 * height                      # up to 16 scalar arguments + the Merkle path
    wit = C.create_string_buffer(cap)
    _check(lib().shp_tee_decode_request(C.c_char_p(body), C.c_size_t(len(body)), C.c_uint32(height), C.byref(kind),
                                        C.byref(circuit), wit, C.c_size_t(cap), C.byref(n)))
    if kind.value != TEE_GENERATE_PROOF:
        return kind.value, None, None
    return kind.value, TEE_CIRCUITS[circuit.value], wit.raw[:32 * n.value]


def tee_encode_proof_response(circuit: str, proof: bytes, instances) -> bytes:
    """Response::EncryptedProof as one JSON frame body (instances: ints or 32-byte LE blobs)"""
    ins = b"".join(x if isinstance(x, (bytes, bytearray)) else int(x).to_bytes(32, "little") for x in instances)
    cap = 256 + 2 * len(proof) + 6 * len(ins) * 2
    out, n = C.create_string_buffer(cap), C.c_size_t()
    _check(lib().shp_tee_encode_proof_response(C.c_int32(TEE_CIRCUITS.index(circuit)), C.c_char_p(proof), C.c_size_t(len(proof)),
                                               C.c_char_p(ins), C.c_size_t(len(ins) // 32), out, C.c_size_t(cap), C.byref(n)))
    return out.raw[:n.value]


def host_cores() -> int:
    """host cores this process can count on: the affinity mask, capped by the cgroup CPU quota"""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 16)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return cores


def prefer_sleeping_waits(lanes: int) -> None:
    """Call before the first Context: with more proving lanes than host cores, stream waits that spin starve the launch
    threads (24 spinning lanes on 16 cores: 870 proofs/s, 1090 sleeping), so ask the library for sleeping waits
    (SHP_BLOCKING_SYNC=1, read by shp_ctx_create) unless the environment already decides"""
    if lanes > host_cores():
        os.environ.setdefault("SHP_BLOCKING_SYNC", "1")


def http_decode_proof_body(body: bytes) -> bytes:
    """GenerateProofPayload {"payload": base64} of the reference's POST /proof -> the payload bytes"""
    n = C.c_size_t()
    _check(lib().shp_http_decode_proof_body(C.c_char_p(body), C.c_size_t(len(body)), None, C.c_size_t(0), C.byref(n)))
    out = C.create_string_buffer(max(1, n.value))
    _check(lib().shp_http_decode_proof_body(C.c_char_p(body), C.c_size_t(len(body)), out, C.c_size_t(n.value), C.byref(n)))
    return out.raw[:n.value]


def tee_decode_payload(payload: bytes, height: int):
    """the Payload JSON of a GenerateProof request -> (circuit name, witness bytes)"""
    circuit, n = C.c_int32(), C.c_size_t()
    cap = 16 * 32 + 32 * 7 * height
    wit = C.create_string_buffer(cap)
    _check(lib().shp_tee_decode_payload(C.c_char_p(payload), C.c_size_t(len(payload)), C.c_uint32(height), C.byref(circuit), wit,
                                        C.c_size_t(cap), C.byref(n)))
    return TEE_CIRCUITS[circuit.value], wit.raw[:32 * n.value]


class ServerOptions(C.Structure):
    """shp_server_options: 0 in a field = its default"""
    _fields_ = [("struct_size", C.c_uint32), ("spawn_timeout_ms", C.c_uint32), ("run_timeout_ms", C.c_uint32),
                ("recv_timeout_ms", C.c_uint32), ("idle_timeout_ms", C.c_uint32), ("max_connections", C.c_uint32),
                ("max_http_body", C.c_uint32)]


class NativeServer:
    """The library's own TCP front of the proving pools (shp_server, csrc/server.hip): the request loop of the reference's
    enclave server (/root/reference/tee/crates/shielder-prover-tee/src/server.rs:69-90) and the HTTP front of its prover
    server (tee/crates/shielder-prover-server/src/main.rs:47-59) on one listener, without an interpreter in the request
    path.  pools: {circuit type name: Pool or [Pool, ...]} -- several pools of one type (one per GPU) share its requests,
    least loaded first; may be empty (Ping / TeePublicKey only).  options: spawn_timeout_ms, run_timeout_ms,
    recv_timeout_ms, idle_timeout_ms, max_connections, max_http_body."""

    def __init__(self, pools: dict, height: int, host: str = "127.0.0.1", port: int = 0, **options):
        flat = [(n, p) for n, ps in pools.items() for p in (ps if isinstance(ps, (list, tuple)) else [ps])]
        arr = (C.c_void_p * max(1, len(flat)))(*[p.h for _, p in flat])
        kinds = (C.c_int32 * max(1, len(flat)))(*[TEE_CIRCUITS.index(n) for n, _ in flat])
        self.pools = pools                               # the pools must outlive the server
        self.h = C.c_void_p()
        opt = ServerOptions(struct_size=C.sizeof(ServerOptions), **options)
        _check(lib().shp_server_start_ex(arr, kinds, C.c_uint32(len(flat)), C.c_uint32(height), C.c_char_p(host.encode()),
                                         C.c_uint16(port), C.byref(opt), C.byref(self.h)))
        p = C.c_uint16()
        _check(lib().shp_server_port(self.h, C.byref(p)))
        self.port = p.value

    def stats(self):
        """(proofs answered, connections ended by an error, connections accepted)"""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib().shp_server_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def timeouts(self):
        """(requests that found every queue full for the spawn timeout, proofs abandoned at the run timeout, frames or
        HTTP requests that stalled for the receive timeout)"""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib().shp_server_timeouts(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def close(self):
        if self.h:
            lib().shp_server_stop(self.h)
            self.h = C.c_void_p()
