"""CPU / GPU affinity codes (reference: accel_rl/scripts/launching/affinities.py:8-180).

Same code strings ("0slt_8gpu_1cxg_1cxr_64cpu_64hto_2skt"), same functions, same resulting
dicts (`gpu`, `gpu_cpus`, `sim_cpus`, `sim_cores`, `gpu_cores`), so launch scripts and run
managers written for the reference keep working.  On the device sampler only `gpu` (which
MI355X the process drives) and `gpu_cpus` (where the master thread may be pinned) matter --
there are no simulation worker processes to pin -- but `sim_*` are still reported for tools
that read them (e.g. the CPU baseline sampler in oracle/).

Layout rule (reference :53-101): physical cores are split evenly over the sockets; on each
socket the first `gpus per socket` cores drive the GPUs, the rest are simulation cores handed
out GPU by GPU and context by context; hyperthread siblings sit `ht_offset` logical CPUs up.
"""

_FIELDS = (("gpu", "n_gpu"), ("cxg", "ctx_per_gpu"), ("cxr", "ctx_per_run"), ("cpu", "n_cpu_cores"),
           ("hto", "ht_offset"), ("skt", "n_socket"))
_SLOT = "slt"


def encode_affinity_params(n_gpu, ctx_per_gpu, ctx_per_run, n_cpu_cores, ht_offset=None, n_socket=None,
                           run_slot=None):
    values = dict(n_gpu=n_gpu, ctx_per_gpu=ctx_per_gpu, ctx_per_run=ctx_per_run, n_cpu_cores=n_cpu_cores,
                  ht_offset=ht_offset, n_socket=n_socket)
    code = "_".join("%d%s" % (values[name], abbrev) for abbrev, name in _FIELDS if values[name] is not None)
    if run_slot is not None:
        assert run_slot <= ctx_per_run // (n_gpu * ctx_per_gpu)
        code = prepend_run_slot_code(run_slot, code)
    return code


def prepend_run_slot_code(run_slot, affinities_code):
    return "%d%s_%s" % (run_slot, _SLOT, affinities_code)


def decode_affinity_params(affinities_code):
    names = dict(_FIELDS)
    out = dict()
    for token in affinities_code.split("_"):
        abbrev, value = token[-3:], int(token[:-3])
        if abbrev not in names:
            raise ValueError("Unrecognized affinity code abbreviation: ", abbrev)
        out[names[abbrev]] = value
    return out


def get_affinities(run_slot_affinities_code):
    slot, code = run_slot_affinities_code.split("_", 1)
    assert slot[-3:] == _SLOT
    return build_affinities(int(slot[:-3]), **decode_affinity_params(code))


def hyperthreads(cpu, ht_offset):
    if isinstance(cpu, list):
        return tuple(cpu + [c + ht_offset for c in cpu])
    return (cpu, cpu + ht_offset)


def build_affinities(run_slot, n_gpu, ctx_per_gpu, ctx_per_run, n_cpu_cores, ht_offset=None, n_socket=1):
    n_ctx = n_gpu * ctx_per_gpu
    assert run_slot < n_ctx // ctx_per_run
    assert n_gpu >= n_socket and n_gpu % n_socket == 0 and n_cpu_cores % n_socket == 0
    ht_offset = n_cpu_cores if ht_offset is None else ht_offset
    gpus_per_socket, cores_per_socket = n_gpu // n_socket, n_cpu_cores // n_socket
    sim_cores_per_gpu = n_cpu_cores // n_gpu - 1
    cores_per_ctx = (n_cpu_cores - n_gpu) // n_ctx
    out = []
    for ctx in range(run_slot * ctx_per_run, (run_slot + 1) * ctx_per_run):
        gpu, ctx_in_gpu = divmod(ctx, ctx_per_gpu)
        socket, gpu_in_socket = divmod(gpu, gpus_per_socket)
        first_core = socket * cores_per_socket
        first_sim = first_core + gpus_per_socket + gpu_in_socket * sim_cores_per_gpu + ctx_in_gpu * cores_per_ctx
        sim_cores = list(range(first_sim, first_sim + cores_per_ctx))
        gpu_cores = [first_core + gpu_in_socket] + (sim_cores if ctx_per_gpu > 1 else [])
        out.append(dict(gpu=gpu, gpu_cpus=hyperthreads(gpu_cores, ht_offset), sim_cpus=hyperthreads(sim_cores, ht_offset),
                        sim_cores=sim_cores, gpu_cores=gpu_cores))
    return out[0] if len(out) == 1 else out


def build_all_affinities(n_gpu, ctx_per_gpu, ctx_per_run, n_cpu_cores, ht_offset=None, n_socket=1):
    return [build_affinities(slot, n_gpu, ctx_per_gpu, ctx_per_run, n_cpu_cores, ht_offset, n_socket)
            for slot in range(n_gpu * ctx_per_gpu // ctx_per_run)]
