"""YAML-driven drivers: the reference's experiment mains over the MI355X-native modules.

    python -m miniweatherml_amd.driver <experiment> <input.yaml> [--max-steps N] [--device cuda:0]
    torchrun --nproc-per-node N -m miniweatherml_amd.driver <experiment> <input.yaml>          # one rank per GPU

experiment            reference main                                                    loop body
supercell_example     experiments/supercell_example/driver.cpp:12-89                    dycore, Kessler, sponge_layer, ColumnNudger
community_benchmark   experiments/community_benchmark/driver.cpp:12-92                  the same, timed as "simulation_loop" (:66,82)
simple_city           experiments/simple_city/driver.cpp:9-88                           horiz. sponge, dycore, sponge_layer(dt,1), averager
inference_ponni       experiments/supercell_kessler_surrogate/inference_ponni.cpp        dycore, NN + Kessler, sponge, nudger
gather_statistics     experiments/supercell_kessler_surrogate/gather_statistics.cpp      dycore, Kessler (+ active-cell ratio), sponge, nudger
generate_micro_data   experiments/supercell_kessler_surrogate/generate_micro_data.cpp    dycore, Kessler (+ training samples), sponge, nudger

The YAML keys are the reference's (sim_time, nens, nx_glob, ny_glob, nz, xlen, ylen, zlen, dt_phys, out_prefix, init_data,
out_freq, enable_gravity, file_per_process; keras_weights_h5 / nn_input_scaling / nn_output_scaling for the surrogate).  The
Keras HDF5 file named by `keras_weights_h5` is read by the library's own reader (mw_h5.cpp = ponni::load_h5_weights); a text export
(`keras_weights_txt`, tools/export_mlp_weights.sh) is accepted too; without either, the reference's shipped weight file
(miniweatherml_amd/data/) is used.
"""
import argparse
import os
import sys
import time

import yaml

EXPERIMENTS = ("supercell_example", "community_benchmark", "simple_city", "inference_ponni", "gather_statistics", "generate_micro_data")


def load_config(path):
    """The `config["key"].as<T>()` reads of the reference mains, with their defaults (driver.cpp:22-38)."""
    with open(path) as f:
        cfg = yaml.safe_load(f)
    if not isinstance(cfg, dict):
        raise ValueError("ERROR: Invalid YAML input file")                      # driver.cpp:25
    out = {}
    for key, typ in (("sim_time", float), ("nx_glob", int), ("ny_glob", int), ("nz", int), ("xlen", float), ("ylen", float),
                     ("zlen", float), ("dt_phys", float), ("out_prefix", str), ("init_data", str), ("out_freq", float)):
        if key not in cfg:
            raise KeyError("ERROR: missing key '%s' in the YAML input file" % key)
        out[key] = typ(cfg[key])
    out["nens"] = int(cfg.get("nens", 1))
    out["enable_gravity"] = bool(cfg.get("enable_gravity", True))
    out["file_per_process"] = bool(cfg.get("file_per_process", False))
    for key in ("keras_weights_h5", "keras_weights_txt", "nn_input_scaling", "nn_output_scaling"):
        if key in cfg:
            out[key] = str(cfg[key])
    out["_dir"] = os.path.dirname(os.path.abspath(path))
    return out


def _distributed(device):
    """One process per GPU when launched by torchrun; returns (nranks, myrank, device)."""
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 1, 0, device
    import torch.distributed as dist
    # dmabuf IPC (the host driver of this pool supports nothing else: without it RCCL between processes fails with
    # "hipIpcGetMemHandle: invalid argument").  The HSA runtime reads it when it initialises -- at the first GPU call, below.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    if device.startswith("cuda"):
        device = "cuda:%d" % local
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if device.startswith("cuda") else "gloo", rank=rank, world_size=world)
    return world, rank, device


def _coupler(cfg, device, nranks, myrank, yaml_path):
    from .coupler import Coupler
    c = Coupler(device)
    c.set_option("out_prefix", cfg["out_prefix"])
    c.set_option("init_data", cfg["init_data"])
    c.set_option("out_freq", cfg["out_freq"])
    c.set_option("enable_gravity", cfg["enable_gravity"])
    c.set_option("file_per_process", cfg["file_per_process"])
    c.distribute_mpi_and_allocate_coupled_state(cfg["nz"], cfg["ny_glob"], cfg["nx_glob"], cfg["nens"], nranks, myrank)
    c.set_grid(cfg["xlen"], cfg["ylen"], cfg["zlen"])
    c.set_option("standalone_input_file", yaml_path)
    return c


def _exchange(dycore, coupler):
    from . import modules
    return modules.install_exchange(dycore, coupler)             # all ranks agree on one transport (RCCL, else torch p2p)


def _time_loop(cfg, dycore, coupler, body, max_steps):
    """while (etime < sim_time) { dtphys = ...; body; etime += dtphys; }   (driver.cpp:66-79)"""
    etime, steps = 0.0, 0
    dtphys = cfg["dt_phys"]
    while etime < cfg["sim_time"] and (max_steps is None or steps < max_steps):
        if cfg["dt_phys"] <= 0.0:
            dtphys = dycore.compute_time_step(coupler)
        if etime + dtphys > cfg["sim_time"]:
            dtphys = cfg["sim_time"] - etime
        body(dtphys, etime)
        etime += dtphys
        steps += 1
    return etime, steps


def run(experiment, yaml_path, max_steps=None, device="cuda:0", quiet=False):
    import torch
    from . import modules
    if experiment not in EXPERIMENTS:
        raise ValueError("unknown experiment %r (one of %s)" % (experiment, ", ".join(EXPERIMENTS)))
    cfg = load_config(yaml_path)
    nranks, myrank, device = _distributed(device)
    coupler = _coupler(cfg, device, nranks, myrank, yaml_path)
    dycore = modules.Dynamics_Euler_Stratified_WenoFV()
    info = {"experiment": experiment, "nranks": nranks}
    t_main = time.perf_counter()
    if experiment == "simple_city":
        horiz_sponge, time_averager = modules.Horizontal_Sponge(), modules.Time_Averager()
        coupler.add_tracer("water_vapor", "water_vapor", True, True)           # simple_city/driver.cpp:55-56
        coupler.get_data_manager_readwrite().get("water_vapor").zero_()
        dycore.init(coupler)
        _exchange(dycore, coupler)
        horiz_sponge.init(coupler, 10, 1.0)
        time_averager.init(coupler)

        def body(dt, etime):                                                   # :72-75
            horiz_sponge.apply(coupler, dt, True, True, False, False)
            dycore.time_step(coupler, dt)
            modules.sponge_layer(coupler, dt, 1)
            time_averager.accumulate(coupler, dt)
        etime, steps = _time_loop(cfg, dycore, coupler, body, max_steps)
        time_averager.finalize(coupler)                                        # :82 -> time_averaged_fields.nc
    else:
        column_nudger = modules.ColumnNudger()
        stats = None
        if experiment == "inference_ponni":
            micro = modules.Microphysics_Kessler_Surrogate()
            base = cfg["_dir"]

            def rel(p):
                return p if p is None or os.path.isabs(p) else os.path.normpath(os.path.join(os.getcwd(), p))
            h5 = rel(cfg.get("keras_weights_h5"))
            kw = dict(weights_txt=rel(cfg.get("keras_weights_txt")), weights_h5=h5 if h5 and os.path.exists(h5) else None)
            for k_yaml, k_arg in (("nn_input_scaling", "in_scaling_txt"), ("nn_output_scaling", "out_scaling_txt")):
                p = rel(cfg.get(k_yaml))
                kw[k_arg] = p if p and os.path.exists(p) else None             # else: the shipped tables
            micro.init(coupler, **kw)
        else:
            micro = modules.Microphysics_Kessler()
            micro.init(coupler)                                                # supercell_example/driver.cpp:58
        dycore.init(coupler)                                                   # :59
        _exchange(dycore, coupler)
        column_nudger.set_column(coupler)                                      # :60
        modules.perturb_temperature(coupler)                                   # :61
        from .coupler import Coupler
        datagen = None
        if experiment == "gather_statistics":
            stats = modules.StatisticsGatherer()
        if experiment == "generate_micro_data":
            datagen = modules.DataGenerator()
            datagen.init(coupler, os.getcwd())                                 # generate_micro_data.cpp:66
            info["samples"] = 0

        def body(dt, etime):                                                   # :73-76
            dycore.time_step(coupler, dt)
            if stats is not None or datagen is not None:
                inp = Coupler(device)
                coupler.clone_into(inp)                                        # gather_statistics.cpp:79-80
                micro.time_step(coupler, dt)
                if stats is not None:
                    stats.gather_micro_statistics(inp, coupler, dt, etime)
                else:
                    info["samples"] += datagen.generate_samples_stencil(inp, coupler, dt, etime)
            else:
                micro.time_step(coupler, dt)
                if experiment == "inference_ponni" and not quiet and coupler.is_mainproc():
                    d = micro.mean_diffs(coupler)                              # microphysics_kessler_ponni.h:266-269
                    print("Relative diff rho_v: %r\nRelative diff rho_c: %r\nRelative diff rho_r: %r\nRelative diff temp : %r" %
                          (d["rho_v"], d["rho_c"], d["rho_r"], d["temp"]), flush=True)
            modules.sponge_layer(coupler, dt)
            column_nudger.nudge_to_column(coupler, dt)
        if experiment == "community_benchmark":                                # timer "simulation_loop", community_benchmark/driver.cpp:66,82
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
        etime, steps = _time_loop(cfg, dycore, coupler, body, max_steps)
        if experiment == "community_benchmark":
            torch.cuda.synchronize(device)
            info["simulation_loop_s"] = time.perf_counter() - t0
        if stats is not None:
            stats.finalize(coupler)
            info["ratio_active"] = stats.ratio(coupler)
    torch.cuda.synchronize(device)
    info.update(etime=etime, steps=steps, main_s=time.perf_counter() - t_main, dycore_etime=dycore.etime, num_out=dycore.num_out)
    if not quiet and coupler.is_mainproc():
        print("driver %s: %d steps, etime %.6f s, wall %.3f s%s" % (experiment, steps, etime, info["main_s"],
              (", simulation_loop %.3f s" % info["simulation_loop_s"]) if "simulation_loop_s" in info else ""), flush=True)
    return coupler, dycore, info


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("experiment", choices=EXPERIMENTS)
    ap.add_argument("yaml")
    ap.add_argument("--max-steps", type=int, default=None)
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args(argv)
    if not os.path.exists(a.yaml):
        print("ERROR: Must pass the input YAML filename as a parameter", file=sys.stderr)       # driver.cpp:21
        return 2
    run(a.experiment, a.yaml, a.max_steps, a.device)
    return 0


if __name__ == "__main__":
    sys.exit(main())
