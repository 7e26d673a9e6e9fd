# JCDFHip.jl — thin ccall glue between JuliaChem.jl and libjcdf_hip.so (include/jcdf.h).
#
# NOT runnable in the build image (no Julia there); it is the reference-side binding a
# JuliaChem maintainer adds (see INTEGRATION.md).  Pure ccall, no CUDA.jl / AMDGPU.jl.
# It provides the two things the reference's informal "GPU operator API" consists of
# (SURVEY.md 8b): an SCFGPUData subtype and a function with the signature of
# df_rhf_fock_build_GPU! (src/rhf/energy/DensityFitting/GPUDF.jl:11-14).
# tests/test_julia_glue_contract.py parses every ccall and the two structs of this file and checks
# names, arities and argument types against include/jcdf.h / the ctypes prototypes.
#
# Wiring (3 edits in JuliaChem):
#   src/shared/Shared.jl          include("GPUData_hip.jl")     -> the struct below
#   src/rhf/energy/SCF.jl:389-393 contraction_mode == "HIP"  ->  gpu_data = SCFGPUData_hip()
#   src/rhf/energy/DensityFitting/DensityFitting.jl:51
#       if scf_options.contraction_mode == "HIP"
#           df_rhf_fock_build_HIP!(scf_data, jeri_engine_thread_df, jeri_engine_thread, basis_sets,
#                                  occupied_orbital_coefficients, iteration, scf_options, H, jc_timing)
# With `"contraction_mode": "HIP"` in keywords.scf, example_scripts/minimal-rhf.jl runs unchanged.

module JCDFHip

using JuliaChem.Shared
using JuliaChem.Shared.JCTC
using LinearAlgebra
using MPI

const libjcdf = get(ENV, "JCDF_HIP_LIB", "libjcdf_hip.so")

# ---- status handling: non-zero status -> error(), the reference's convention (GPUDF.jl:39-41)
struct JCDFTimings                      # mirrors jcdf_timings (include/jcdf.h), seconds
    non_zero_coeff_time::Cdouble
    W_time::Cdouble
    K_time::Cdouble
    V_time::Cdouble
    J_time::Cdouble
    density_time::Cdouble
    H_add_time::Cdouble
    copy_J_time::Cdouble
    fock_time::Cdouble
    copy_time::Cdouble
end

struct JCDFGroupTimings                 # mirrors jcdf_group_timings (include/jcdf.h), seconds
    bcast_time::Cdouble
    build_time::Cdouble
    reduce_time::Cdouble
    d2h_time::Cdouble
    total_time::Cdouble
end

function check(h::Ptr{Cvoid}, rc::Int32)
    if rc != 0
        msg = unsafe_string(ccall((:jcdf_last_error, libjcdf), Cstring, (Ptr{Cvoid},), h))
        error("libjcdf_hip status $rc: $msg")
    end
end

function check_group(g::Ptr{Cvoid}, rc::Int32)
    if rc != 0
        msg = unsafe_string(ccall((:jcdf_group_last_error, libjcdf), Cstring, (Ptr{Cvoid},), g))
        error("libjcdf_hip group status $rc: $msg")
    end
end

# ---- the SCFGPUData subtype (counterpart of SCFGPUData_cuda, shared/GPUData_cuda.jl:4-38)
mutable struct SCFGPUData_hip <: SCFGPUData
    handle::Ptr{Cvoid}                          # num_devices == 1: the one jcdf handle
    group::Ptr{Cvoid}                           # num_devices  > 1: a jcdf_group over this rank's devices (F summed on the devices)
    device_Q_range_lengths::Vector{Int}
    device_Q_indices::Vector{UnitRange{Int}}
    number_of_devices_used::Int
    host_fock::Matrix{Float64}                  # ONE host Fock matrix per rank, whatever the device count
    SCFGPUData_hip() = new(C_NULL, C_NULL, Int[], UnitRange{Int}[], 0, zeros(Float64, 0, 0))
end

function destroy!(gd::SCFGPUData_hip)
    gd.handle != C_NULL && ccall((:jcdf_destroy, libjcdf), Int32, (Ptr{Cvoid},), gd.handle)
    gd.group != C_NULL && ccall((:jcdf_group_destroy, libjcdf), Int32, (Ptr{Cvoid},), gd.group)
    gd.handle = C_NULL; gd.group = C_NULL
end

# run_gpu_fock_build! (DensityFitting.jl:78-90): the UNSCREENED ("dense") algorithm when the user forces it
# (df_force_dense / contraction_mode "denseGPU") or, adaptively, for small systems on a single rank — there the reference
# never Schwarz-screens the pairs (DenseGPUDF.jl:9-21), so neither may this path: screening changes the energy by ~1e-6 Eh.
function use_dense_map(scf_data, scf_options, rank, n_ranks)
    df_force_dense = scf_options.df_force_dense || scf_options.contraction_mode == "denseGPU"
    return df_force_dense || (scf_options.df_use_adaptive && scf_data.μ < 800 && rank == 0 && n_ranks == 1)
end

# ---- setup: iteration == 1 branch of df_rhf_fock_build_GPU! (GPUDF.jl:37-165) / df_rhf_fock_build_dense_GPU! (DenseGPUDF.jl:29-72)
function setup!(scf_data, jeri_engine_thread_df, jeri_engine_thread, basis_sets, scf_options, H, jc_timing)
    comm = MPI.COMM_WORLD
    rank = MPI.Comm_rank(comm); n_ranks = MPI.Comm_size(comm)
    num_devices = scf_options.num_devices
    gd = scf_data.gpu_data::SCFGPUData_hip
    N = scf_data.μ; A = scf_data.A; occ = scf_data.occ
    dense = use_dense_map(scf_data, scf_options, rank, n_ranks)
    if dense && n_ranks > 1                     # DenseGPUDF.jl:17-21: "Dense GPU algorithm only supports 1 rank runs"
        error("contraction_mode HIP: the dense (unscreened) algorithm runs on one rank; use the screened path with MPI")
    end

    # host integrals exactly as the reference does them (they stay on the host)
    two_center_integrals = calculate_two_center_intgrals(jeri_engine_thread_df, basis_sets, scf_options)   # GPUDF.jl:43, DenseGPUDF.jl:166
    if dense
        setup_unscreened_screening_matricies(basis_sets, scf_data)                                          # DenseGPUDF.jl:218, SchwarzScreening.jl:97-111
        P = N * N
        pq_p = Ptr{Int64}(C_NULL); pq_q = Ptr{Int64}(C_NULL)       # NULL, NULL with P == N^2: the library's unscreened map c = q + N p
    else
        get_screening_metadata!(scf_data, scf_options.df_screening_sigma, jeri_engine_thread,
                                two_center_integrals, basis_sets, jc_timing)                                # GPUDF.jl:45
        sd = scf_data.screening_data
        P = sd.screened_indices_count
        # inverse of sparse_pq_index_map (1-based Julia -> 0-based C), what GPUDF.jl:422-438 builds on the device
        pq_p = Vector{Int64}(undef, P); pq_q = Vector{Int64}(undef, P)
        for pp in 1:N, qq in 1:N
            idx = sd.sparse_pq_index_map[qq, pp]
            if idx != 0
                pq_p[idx] = pp - 1; pq_q[idx] = qq - 1
            end
        end
    end

    # aux shards: one per device (global device id = rank * num_devices + dev), the reference's own rule
    device_Q_indices, _, device_Q_range_lengths, _ =
        calculate_device_ranges_GPU(scf_data, num_devices, n_ranks, basis_sets)                             # GPUDF.jl:1026-1056
    gd.device_Q_indices = device_Q_indices; gd.device_Q_range_lengths = device_Q_range_lengths
    gd.number_of_devices_used = num_devices
    gd.host_fock = zeros(Float64, N, N)
    # scf flag df_exchange_screen (SCFOptions.jl:92-93; ScreenedDF.jl:431-447, 459-545): K blocks of width N / n_blocks
    # without a kept pair are not computed; n_blocks = df_exchange_n_blocks or the screened mode's default of 10
    xs_blocks = (scf_options.df_screen_exchange && !dense) ?
        (scf_options.df_exchange_n_blocks == 0 ? 10 : scf_options.df_exchange_n_blocks) : 0
    my = [device_Q_indices[dev + rank * num_devices] for dev in 1:num_devices]

    if num_devices == 1
        rows = my[1]
        href = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:jcdf_create, libjcdf), Int32, (Ref{Ptr{Cvoid}}, Int32), href, 0)
        rc == 0 || error(unsafe_string(ccall((:jcdf_last_error, libjcdf), Cstring, (Ptr{Cvoid},), C_NULL)))
        h = href[]
        gd.handle = h
        check(h, ccall((:jcdf_set_exchange_screening, libjcdf), Int32, (Ptr{Cvoid}, Int64), h, xs_blocks))
        check(h, ccall((:jcdf_configure, libjcdf), Int32,
                       (Ptr{Cvoid}, Int64, Int64, Int64, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}),
                       h, N, A, first(rows) - 1, last(rows), occ, P, pq_p, pq_q))
        # L = chol((P|Q)) and L^-1 on the device (the placement of CUSOLVER.potrf!/trtri! in DenseGPUDF.jl:185-193; the
        # screened path does them with host LAPACK at GPUDF.jl:890-891)
        check(h, ccall((:jcdf_set_metric, libjcdf), Int32, (Ptr{Cvoid}, Ptr{Float64}), h, two_center_integrals))
        check(h, ccall((:jcdf_set_core_hamiltonian, libjcdf), Int32, (Ptr{Cvoid}, Ptr{Float64}),
                       h, rank == 0 ? H : C_NULL))                                                          # GPUDF.jl:158-161
    else
        # all devices of this rank behind ONE group: C_occ up once, F summed on the devices (RCCL reduce-scatter over xGMI,
        # or the library's peer-mapped slice sums), F down once — replaces the task / H2D / D2H per device and the host
        # axpy! of GPUDF.jl:188-193, 206, 267-277
        gref = Ref{Ptr{Cvoid}}(C_NULL)
        device_ids = Int32[dev - 1 for dev in 1:num_devices]
        rc = ccall((:jcdf_group_create, libjcdf), Int32, (Ref{Ptr{Cvoid}}, Int32, Ptr{Int32}), gref, num_devices, device_ids)
        rc == 0 || error(unsafe_string(ccall((:jcdf_group_last_error, libjcdf), Cstring, (Ptr{Cvoid},), C_NULL)))
        g = gref[]
        gd.group = g
        shard_q0 = Int64[[first(r) - 1 for r in my]; last(my[end])]                # num_devices + 1 entries, 0-based
        check_group(g, ccall((:jcdf_group_set_exchange_screening, libjcdf), Int32, (Ptr{Cvoid}, Int64), g, xs_blocks))
        check_group(g, ccall((:jcdf_group_configure, libjcdf), Int32,
                             (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Int64, Int64, Ptr{Int64}, Ptr{Int64}),
                             g, N, A, shard_q0, occ, P, pq_p, pq_q))
        check_group(g, ccall((:jcdf_group_set_metric, libjcdf), Int32, (Ptr{Cvoid}, Ptr{Float64}), g, two_center_integrals))
        check_group(g, ccall((:jcdf_group_set_core_hamiltonian, libjcdf), Int32, (Ptr{Cvoid}, Ptr{Float64}),
                             g, rank == 0 ? H : C_NULL))                                                    # member 0 only
    end

    # three-centre blocks: block gid is computed by its owner (GPUDF.jl:51-57; dense: DenseGPUDF.jl:198-200, 248-256),
    # broadcast across ranks, and pushed; the library skips blocks above the diagonal of L^-1 and, in a group, uploads a
    # block once and lets the other devices fetch it device-to-device.
    for gid in 1:(num_devices * n_ranks)
        owner = (gid - 1) ÷ num_devices
        rows = device_Q_indices[gid]
        T = if owner != rank
            zeros(Float64, length(rows), P)
        elseif dense && num_devices == 1
            reshape(calculate_three_center_integrals(jeri_engine_thread_df, basis_sets, scf_options, scf_data, 0, 1, false, false),
                    (A, N * N))                                                                             # (A, N, N) -> (rows, N^2)
        else
            calculate_three_center_integrals(jeri_engine_thread_df, basis_sets, scf_options, scf_data, gid - 1,
                                             num_devices * n_ranks, true)   # packed (rows, P); with the unscreened map P = N^2
        end
        n_ranks > 1 && MPI.Bcast!(T, owner, comm)
        if num_devices == 1
            check(gd.handle, ccall((:jcdf_push_three_center, libjcdf), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}),
                                   gd.handle, first(rows) - 1, last(rows), T))
        else
            check_group(gd.group, ccall((:jcdf_group_push_three_center, libjcdf), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}),
                                        gd.group, first(rows) - 1, last(rows), T))
        end
    end
    jc_timing.non_timing_data[JCTC.contraction_algorithm] = dense ? "dense hip" : "screened hip"
    jc_timing.non_timing_data[JCTC.GPU_num_devices] = string(num_devices)
    if num_devices > 1
        jc_timing.non_timing_data["GPU_reduce_transport"] =
            unsafe_string(ccall((:jcdf_group_transport, libjcdf), Cstring, (Ptr{Cvoid},), gd.group))
    end
end

# ---- the operator: same signature and side effects as df_rhf_fock_build_GPU! (GPUDF.jl:11-14)
function df_rhf_fock_build_HIP!(scf_data, jeri_engine_thread_df, jeri_engine_thread, basis_sets,
                                occupied_orbital_coefficients::Matrix{Float64}, iteration::Int,
                                scf_options, H::Matrix{Float64}, jc_timing)
    gd = scf_data.gpu_data::SCFGPUData_hip
    iteration == 1 && setup!(scf_data, jeri_engine_thread_df, jeri_engine_thread, basis_sets, scf_options, H, jc_timing)
    num_devices = gd.number_of_devices_used
    times = Vector{JCDFTimings}(undef, num_devices)
    fock_copy_time = 0.0
    total = @elapsed begin
        if num_devices == 1
            t = Ref{JCDFTimings}()
            check(gd.handle, ccall((:jcdf_fock_build, libjcdf), Int32,
                                   (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ref{JCDFTimings}),
                                   gd.handle, occupied_orbital_coefficients, gd.host_fock, t))   # C_occ is (N, occ) column-major
            times[1] = t[]
            fock_copy_time = t[].copy_time
        else
            # one call drives every device (GPUDF.jl:189-193 spawns a task per device): C_occ H2D once + peer copies, all shards
            # concurrently, reduce-scatter of F on the devices, one pass of D2H into host_fock — no host axpy! (GPUDF.jl:267-277)
            gt = Ref{JCDFGroupTimings}()
            check_group(gd.group, ccall((:jcdf_group_fock_build, libjcdf), Int32,
                                        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{JCDFTimings}, Ref{JCDFGroupTimings}),
                                        gd.group, occupied_orbital_coefficients, gd.host_fock, times, gt))
            fock_copy_time = gt[].bcast_time + gt[].reduce_time + gt[].d2h_time
        end
    end
    scf_data.two_electron_fock = gd.host_fock
    # every key df_rhf_fock_build_GPU! writes (GPUDF.jl:280-301); the reference's analysis scripts read them all.
    # Steps this library fuses away report 0: V comes out of the W pass, no density is formed, the symmetrisation and
    # the H addition are part of the assemble launch (reported under gpu_copy_J_time).
    for (dev, t) in enumerate(times)
        jc_timing.timings[JCTiming_GPUkey(JCTC.GPU_W_time, dev, iteration)] = t.W_time
        jc_timing.timings[JCTiming_GPUkey(JCTC.GPU_V_time, dev, iteration)] = t.V_time
        jc_timing.timings[JCTiming_GPUkey(JCTC.GPU_J_time, dev, iteration)] = t.J_time
        jc_timing.timings[JCTiming_GPUkey(JCTC.GPU_K_time, dev, iteration)] = t.K_time
        jc_timing.timings[JCTiming_GPUkey(JCTC.GPU_density_time, dev, iteration)] = t.density_time
        jc_timing.timings[JCTiming_GPUkey(JCTC.gpu_fock_time, dev, iteration)] = t.fock_time
        jc_timing.timings[JCTiming_GPUkey(JCTC.GPU_non_zero_coeff_time, dev, iteration)] = t.non_zero_coeff_time
        jc_timing.timings[JCTiming_GPUkey(JCTC.gpu_copy_J_time, dev, iteration)] = t.copy_J_time
        jc_timing.timings[JCTiming_GPUkey(JCTC.gpu_copy_sym_time, dev, iteration)] = 0.0
    end
    jc_timing.timings[JCTiming_key(JCTC.K_time, iteration)] = maximum(t.K_time for t in times)
    jc_timing.timings[JCTiming_key(JCTC.W_time, iteration)] = maximum(t.W_time for t in times)
    jc_timing.timings[JCTiming_key(JCTC.V_time, iteration)] = maximum(t.V_time for t in times)
    jc_timing.timings[JCTiming_key(JCTC.J_time, iteration)] = maximum(t.J_time for t in times)
    jc_timing.timings[JCTiming_key(JCTC.fock_time, iteration)] = maximum(t.fock_time for t in times)
    jc_timing.timings[JCTiming_GPUkey(JCTC.GPU_H_add_time, 1, iteration)] = times[1].H_add_time
    jc_timing.timings[JCTiming_key(JCTC.fock_gpu_cpu_copy_reduce_time, iteration)] = fock_copy_time
    jc_timing.timings[JCTiming_key(JCTC.total_fock_gpu_time, iteration)] = total
    return nothing       # ranks are summed by the caller's MPI.Allreduce! (DensityFitting.jl:68-71)
end

export SCFGPUData_hip, df_rhf_fock_build_HIP!, destroy!

end # module
