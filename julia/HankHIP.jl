# HankHIP.jl — the reference-side binding a maintainer of vasudeva-ram/Julia-NewtonRaphsonHANK would add:
# the SAME Julia signatures as BackwardIteration.jl / ForwardIteration.jl, bodies replaced by `ccall`s
# into libhank_hip.so (include/hank_hip.h). `include` it AFTER the reference's own files; NewtonRaphson.jl,
# ModelParser.jl, SteadyState.jl and the YAML stay untouched.
#
# NOT EXERCISED IN THIS REPOSITORY'S CI: no Julia toolchain exists in the build image (SURVEY.md §8c).
# The C ABI it binds is exercised through the Python ctypes binding (julia-newtonraphsonhank_amd/hip.py),
# which makes exactly the same calls.
#
#   policy_seqs = BackwardIteration(x, exog_paths, mod, ss_end)          # BackwardIteration.jl:46-49
#   agg_seqs    = ForwardIteration(policy_seqs, mod, ss_initial)         # ForwardIteration.jl:253-255
#
# With x::Vector{Float64} both run the fused Float64 sweep (hank_primal); with
# x::Vector{ForwardDiff.Dual{T,Float64,N}} the N partials travel as one tangent batch (hank_jvp), so
# `JVP(fullFunction, x, y)` (GeneralStructures.jl:542-550) and ForwardDiff.jacobian chunks work unchanged.

using ForwardDiff: Dual, Partials, value, partials, tagtype

const LIBHANK = get(ENV, "HANK_HIP_LIB", joinpath(@__DIR__, "..", "julia-newtonraphsonhank_amd", "libhank_hip.so"))

struct HankModelC            # mirrors `hank_model` in include/hank_hip.h
    n_a::Int32; n_e::Int32; T::Int32; value_fn_id::Int32
    a_grid::Ptr{Float64}; z_grid::Ptr{Float64}; Pi::Ptr{Float64}
    beta::Float64; gamma::Float64; borrow_cons::Float64
end

mutable struct HankCtx
    ptr::Ptr{Cvoid}
    P::Int; G::Int; n_a::Int; n_e::Int
    hh_rows::Tuple        # names of the xVals rows the native family reads, in the order the library expects
    outputs::Tuple        # the heterogeneous variables the family returns, in the order of hank_get_het_outputs
end

# native kernel families by value-function name: id (include/hank_hip.h), the household inputs it reads, and the heterogeneous
# variables it returns (output 1 = the policy variable of the endogenous dimension, output 2 = consumption, the c_grid of
# KrusellSmith.jl:79 returned as a second policy: hank_get_het_outputs)
const _FAMILIES = Dict(:ValueFunction => (0, (:r, :w), (:KD, :C)),                # KrusellSmith.jl:43-83, :53-54
                       :HANKValueFunction => (1, (:r, :om, :Tr), (:A, :C)))      # one-asset HANK (not in the reference)

# one device context per (SequenceModel, HIP device); device -1 = the current one. Keyed on the model ITSELF (an IdDict keeps it
# alive: an objectid can be recycled by a later model and hand it a context built for other grids) and guarded by a lock:
# sharded_jvp_columns reaches it from several tasks.
const _CTX = IdDict{Any,Dict{Int,HankCtx}}()
const _CTX_LOCK = ReentrantLock()

function _check(ctx::Ptr{Cvoid}, rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:hank_last_error, LIBHANK), Cstring, (Ptr{Cvoid},), ctx))
    error(msg)      # the library's messages restate the reference's own exceptions (knots / DomainError)
end

# `device`: HIP device ordinal (hank_create_on) — one context per GPU of a node lets ONE Julia process shard the columns of
# a tangent batch over the GPUs (sharded_jvp_columns below); `nothing` = the calling thread's current device (hank_create)
function hank_context(model::SequenceModel; device::Union{Nothing,Integer} = nothing)
    lock(_CTX_LOCK) do
    get!(get!(() -> Dict{Int,HankCtx}(), _CTX, model), device === nothing ? -1 : Int(device)) do
        w = model.heterogeneity.wealth; p = model.heterogeneity.productivity
        a = collect(Float64, w.grid); z = collect(Float64, p.grid); Π = Matrix{Float64}(p.transition)
        haskey(_FAMILIES, nameof(model.value_fn)) || error("no native kernel family for $(model.value_fn)")
        fam_id, rows, outs = _FAMILIES[nameof(model.value_fn)]
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve a z Π begin
            m = HankModelC(w.n, p.n, model.compspec.T, fam_id, pointer(a), pointer(z), pointer(Π),
                           model.params.β, model.params.γ, model.params.borrow_cons)
            rc = device === nothing ?
                ccall((:hank_create, LIBHANK), Cint, (Ref{HankModelC}, Ref{Ptr{Cvoid}}), m, ref) :
                ccall((:hank_create_on, LIBHANK), Cint, (Ref{HankModelC}, Int32, Ref{Ptr{Cvoid}}), m, Int32(device), ref)
        end
        _check(ref[], rc)
        ctx = HankCtx(ref[], model.compspec.T - 1, w.n * p.n, w.n, p.n, rows, outs)
        @assert ccall((:hank_n_hh, LIBHANK), Cint, (Ptr{Cvoid},), ref[]) == length(rows)
        finalizer(c -> ccall((:hank_destroy, LIBHANK), Cint, (Ptr{Cvoid},), c.ptr), ctx)
        ctx
    end
    end
end

# rows of xVals the value function reads (KS: r_t, w_t, KrusellSmith.jl:53-54)
function _household_inputs(xVec_endog, model)
    @unpack T, n_endog = model.compspec
    xMat = reshape(xVec_endog, n_endog, T - 1)
    ek = vars_of_type(model, :endogenous)
    rows = [findfirst(==(k), ek) for k in hank_context(model).hh_rows]
    return xMat[rows, :]                                      # n_hh x (T-1), eltype of x
end

# What BackwardIteration returns: the reference's NamedTuple-of-Vector{Matrix} surface (seqs.KD[t]),
# copied from HBM on demand. The reference calls BackwardIteration with FOUR positionals
# (NewtonRaphson.jl:78) — `ss_initial` only reaches ForwardIteration (:79) — and the fused device sweep
# needs D_0, so the device call is DEFERRED: ForwardIteration runs the one fused sweep; reading a policy
# matrix before that runs it with a placeholder D_0 (policies do not depend on D_0).
mutable struct DevicePolicySeqs{TF}
    ctx::HankCtx; het_keys::Tuple; N::Int
    xhh::Matrix{Float64}; dxhh::Union{Nothing,Array{Float64,3}}; value::Matrix{Float64}
    D0::Union{Nothing,Vector{Float64}}; agg::Union{Nothing,Vector{Float64}}; dagg::Union{Nothing,Matrix{Float64}}
    aggs::Union{Nothing,Matrix{Float64}}; daggs::Union{Nothing,Array{Float64,3}}      # (P, n_het), (P, n_het, N): models with two heterogeneous variables
    grids::Tuple{Vector{Float64},Vector{Float64}}                                      # wealth and productivity grids (consumption policy)
end

# a Dual of the caller's own type TF = Dual{Tag,Float64,N} from a value and N partials: the inner
# constructor takes a Partials (ForwardDiff.jl/src/dual.jl:14-21); Dual{Tag}(v, ::Partials) is the
# public form (:59-66). There is no TF(value, p1, ..., pN) method.
@inline _mkdual(::Type{TF}, v::Float64, p::NTuple{N,Float64}) where {N,TF<:Dual} = Dual{tagtype(TF)}(v, Partials(p))

function _run_block!(s::DevicePolicySeqs, D0::Vector{Float64})
    ctx = s.ctx
    _check(ctx.ptr, ccall((:hank_set_boundary, LIBHANK), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), ctx.ptr, s.value, D0))
    agg = Vector{Float64}(undef, ctx.P)
    dagg = nothing
    if s.dxhh === nothing
        _check(ctx.ptr, ccall((:hank_primal, LIBHANK), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), ctx.ptr, s.xhh, agg))
    else      # a Dual pass carries value and partials together (NewtonRaphson.jl:95): one dual-sweep call
        dagg = Matrix{Float64}(undef, ctx.P, s.N)
        _check(ctx.ptr, ccall((:hank_primal_jvp, LIBHANK), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}),
                              ctx.ptr, s.xhh, s.dxhh, s.N, agg, dagg))
    end
    s.D0, s.agg, s.dagg = D0, agg, dagg
    if length(s.het_keys) > 1     # every heterogeneous variable's aggregate, reduced by the same sweeps (ForwardIteration.jl:303-307)
        n_het = length(ctx.outputs)
        s.aggs = Matrix{Float64}(undef, ctx.P, n_het)
        s.daggs = s.dxhh === nothing ? nothing : Array{Float64}(undef, ctx.P, n_het, s.N)
        _check(ctx.ptr, ccall((:hank_get_het_outputs, LIBHANK), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}),
                              ctx.ptr, Int32(n_het), s.dxhh === nothing ? C_NULL : s.dxhh, Int32(s.N), s.aggs, s.daggs === nothing ? C_NULL : s.daggs))
    end
    return s
end

# same positional signature as BackwardIteration.jl:46-49. No device work happens here (see above);
# `ss_initial` (optional keyword, not in the reference) runs the fused sweep straight away.
function BackwardIteration(xVec_endog, exog_paths::NamedTuple, model::SequenceModel, ss_end; ss_initial = nothing)
    ctx = hank_context(model)
    TF = eltype(xVec_endog)
    xd = _household_inputs(xVec_endog, model)
    xhh = Matrix{Float64}(value.(xd))
    N = TF <: Dual ? length(partials(first(xVec_endog))) : 0
    dxhh = N > 0 ? Float64[partials(xd[k, t])[n] for k in 1:size(xd, 1), t in 1:ctx.P, n in 1:N] : nothing   # (n_hh, P, N)
    s = DevicePolicySeqs{TF}(ctx, vars_of_type(model, :heterogeneous), N, xhh, dxhh, Matrix{Float64}(ss_end.value),
                             nothing, nothing, nothing, nothing, nothing,
                             (collect(Float64, model.heterogeneity.wealth.grid), collect(Float64, model.heterogeneity.productivity.grid)))
    ss_initial === nothing || _run_block!(s, Vector{Float64}(ss_initial.D))
    return s
end

# seqs.KD -> Vector of T-1 (Dual) matrices, as in the reference (BackwardIteration.jl:110-115)
function Base.getproperty(s::DevicePolicySeqs{TF}, k::Symbol) where {TF}
    k in fieldnames(DevicePolicySeqs) && return getfield(s, k)
    ctx = getfield(s, :ctx); N = getfield(s, :N)
    getfield(s, :agg) === nothing && _run_block!(s, fill(1.0 / ctx.G, ctx.G))     # read before ForwardIteration
    pol = Array{Float64}(undef, ctx.n_a, ctx.n_e, ctx.P)
    _check(ctx.ptr, ccall((:hank_get_policy_seq, LIBHANK), Cint, (Ptr{Cvoid}, Ptr{Float64}), ctx.ptr, pol))
    seq = if N == 0
        [pol[:, :, t] for t in 1:ctx.P]
    else
        dpol = Array{Float64}(undef, ctx.n_a, ctx.n_e, ctx.P, N)
        _check(ctx.ptr, ccall((:hank_get_dpolicy_seq, LIBHANK), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), ctx.ptr, N, dpol))
        [[_mkdual(TF, pol[a, e, t], ntuple(n -> dpol[a, e, t, n], N)) for a in 1:ctx.n_a, e in 1:ctx.n_e] for t in 1:ctx.P]
    end
    k == ctx.outputs[1] && return seq
    k == :C || error("the native family returns $(ctx.outputs), not :$k")
    # consumption, the budget residual of KrusellSmith.jl:79: (1 + r) .* policy_a .+ (w .* labor_mat [.+ tr]) .- griddedpolicy
    xhh = getfield(s, :xhh); dxhh = getfield(s, :dxhh)
    xin(j, t) = N == 0 ? xhh[j, t] : _mkdual(TF, xhh[j, t], ntuple(n -> dxhh[j, t, n], N))
    grid = getfield(s, :grids)[1]; z = getfield(s, :grids)[2]
    return [[(1 + xin(1, t)) * grid[a] + (xin(2, t) * z[e] + (size(xhh, 1) > 2 ? xin(3, t) : 0.0)) - seq[t][a, e]
             for a in 1:ctx.n_a, e in 1:ctx.n_e] for t in 1:ctx.P]
end

# same signature as ForwardIteration.jl:253-255 for sequences that came from BackwardIteration above:
# this is where the ONE fused sweep of a fullFunction evaluation runs (NewtonRaphson.jl:78-79)
function ForwardIteration(seqs::DevicePolicySeqs{TF}, model::SequenceModel, ss_initial) where {TF}
    all(k -> k in seqs.ctx.outputs, seqs.het_keys) || error("the native family returns $(seqs.ctx.outputs) (got $(seqs.het_keys))")
    D0 = Vector{Float64}(ss_initial.D)
    (seqs.agg === nothing || D0 != seqs.D0) && _run_block!(seqs, D0)
    if length(seqs.het_keys) == 1
        agg, dagg = seqs.agg, seqs.dagg
        out = seqs.N == 0 ? agg : [_mkdual(TF, agg[t], ntuple(n -> dagg[t, n], seqs.N)) for t in 1:length(agg)]
        return NamedTuple{seqs.het_keys}((out,))
    end
    # one aggregate per heterogeneous variable, each its own policy dotted with the same D_t (ForwardIteration.jl:303-307)
    aggs, daggs = seqs.aggs, seqs.daggs
    col(k) = findfirst(==(k), seqs.ctx.outputs)
    outs = map(seqs.het_keys) do k
        j = col(k)
        seqs.N == 0 ? aggs[:, j] : [_mkdual(TF, aggs[t, j], ntuple(n -> daggs[t, j, n], seqs.N)) for t in 1:size(aggs, 1)]
    end
    return NamedTuple{seqs.het_keys}(Tuple(outs))
end

# ---- one process, several GPUs (GeneralStructures.jl:542-550: JVP is linear in `tangent`, so tangent columns shard) ----------
# dagg = J_household(x) * dxhh, the (P, N) household-block part of N JVPs at once, columns split over `devices`. Every
# context replays the (cheap) Float64 sweep; GPU g takes the contiguous column block g; each task blocks in its own
# `hank_primal_jvp` while the other GPUs run; the blocks land in ONE host matrix — no collective. This is what replaces the
# column loop of `getSteadyStateJacobian` (SteadyStateJacobian.jl:240-243) or a `ForwardDiff.jacobian` chunk loop.
function sharded_jvp_columns(model::SequenceModel, ss_end, ss_initial, xhh::Matrix{Float64}, dxhh::Array{Float64,3}; devices = [0])
    P, N, W = size(dxhh, 2), size(dxhh, 3), length(devices)
    dagg = Matrix{Float64}(undef, P, N)
    base, extra = divrem(N, W)
    value = Matrix{Float64}(ss_end.value); D0 = Vector{Float64}(ss_initial.D)
    ctxs = [hank_context(model; device = dev) for dev in devices]      # resolved (or created) on the calling task, before any task is spawned
    # (the blocking ccalls only overlap when Julia runs at least length(devices) threads: `julia -t N`; with one thread the tasks
    # serialise — then drive the contexts through the asynchronous hank_*_dev entries instead)
    @sync for (g, dev) in enumerate(devices)
        lo = (g - 1) * base + min(g - 1, extra) + 1
        hi = lo + base + (g <= extra ? 1 : 0) - 1
        hi < lo && continue
        Threads.@spawn begin                     # a context is used by ONE task at a time; different contexts run concurrently
            ctx = ctxs[g]
            _check(ctx.ptr, ccall((:hank_set_boundary, LIBHANK), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), ctx.ptr, value, D0))
            agg = Vector{Float64}(undef, P)
            blk = dxhh[:, :, lo:hi]
            out = Matrix{Float64}(undef, P, hi - lo + 1)
            _check(ctx.ptr, ccall((:hank_primal_jvp, LIBHANK), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}),
                                  ctx.ptr, xhh, blk, Int32(hi - lo + 1), agg, out))
            dagg[:, lo:hi] = out
        end
    end
    return dagg
end

# The same with the blocks left in HBM and assembled on ONE GPU over xGMI (hank_gather_columns): `d_blocks[k]` is the device pointer
# of context k's (P, N_k[k]) block as hank_jvp_dev / hank_primal_jvp_dev wrote it, `d_out` a (P, sum N_k) buffer on ctxs[1]'s device.
# Asynchronous: hank_sync(ctxs[1]) — or work on its stream — sees the assembled matrix.
function gather_columns!(ctxs::Vector{HankCtx}, d_blocks::Vector{Ptr{Float64}}, N_k::Vector{Int32}, d_out::Ptr{Float64})
    ptrs = Ptr{Cvoid}[c.ptr for c in ctxs]
    _check(ctxs[1].ptr, ccall((:hank_gather_columns, LIBHANK), Cint, (Ptr{Ptr{Cvoid}}, Int32, Ptr{Ptr{Float64}}, Ptr{Int32}, Ptr{Float64}),
                              ptrs, Int32(length(ctxs)), d_blocks, N_k, d_out))
    return d_out
end

# which kernel family ran the last tangent sweep and how the context is configured (hank_info): (last_tangent_family, wide_mode,
# wide_min, wide_supported, xjvp_max, record_diet, record_bytes, 0); families: 0 per-period launches, 1 XCD-persistent, 2 on-chip wide
function hank_info(ctx::HankCtx)
    out = zeros(Int64, 8)
    _check(ctx.ptr, ccall((:hank_info, LIBHANK), Cint, (Ptr{Cvoid}, Ptr{Int64}), ctx.ptr, out))
    return out
end

# ---- the household block of getSteadyStateJacobian (SteadyStateJacobian.jl:187-256, :293-323, :358-387) ------------------------
# JBI (n_endog forward-mode JVPs through BackwardIteration seeded at the last period, :240-243), JFI (Zygote pullbacks
# through ForwardIteration, :249-253) and helper = JFI * JBI (:300-305) collapse into ONE call: hank_fake_news returns the
# fake-news matrix F[u, j, k] and the direct term Dv[j, k] of the household block at the steady state; the Toeplitz
# recursion (:363-371) stays here. Returns J[t, s, k] = d agg_t / d xhh_{k,s} (xhh rows in the order of `ctx.hh_rows`);
# the caller places it behind the equations' direct blocks (:124-145) exactly where `helper` goes today.
function household_jacobian_toeplitz(model::SequenceModel, ss; device = nothing)
    ctx = hank_context(model; device = device)
    P = ctx.P; n_hh = length(ctx.hh_rows)
    xhh = Float64[ss.vars[k] for k in ctx.hh_rows, _ in 1:P]                      # the constant steady-state path (:53-57)
    _check(ctx.ptr, ccall((:hank_set_boundary, LIBHANK), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), ctx.ptr,
                          Matrix{Float64}(ss.value), Vector{Float64}(ss.D)))
    agg = Vector{Float64}(undef, P)
    _check(ctx.ptr, ccall((:hank_primal, LIBHANK), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), ctx.ptr, xhh, agg))
    F = Array{Float64}(undef, P, P, n_hh); Dv = Matrix{Float64}(undef, P, n_hh)
    _check(ctx.ptr, ccall((:hank_fake_news, LIBHANK), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), ctx.ptr, F, Dv))
    J = similar(F)
    for k in 1:n_hh
        J[1, :, k] = Dv[:, k] .+ F[1, :, k]
        for t in 2:P
            J[t, 1, k] = F[t, 1, k]
            J[t, 2:P, k] = J[t-1, 1:P-1, k] .+ F[t, 2:P, k]
        end
    end
    return J
end
