# AGPHip.jl -- Julia shim over libagp_hip.so (include/agp_hip.h).
#
# NOT EXECUTED IN THE BUILD ENVIRONMENT (no julia binary there or on the GPU box); it is the binding a maintainer of
# AugmentedGaussianProcesses.jl would add so that SVGP / MOSVGP models with AnalyticVI / AnalyticSVI run their hot path on an
# MI355X, written against the reference's own internals (file:line cited at every method).  AMDGPU.jl is used ONLY for device /
# stream / buffer handles (ROCArray, the stream pointer); there is no KernelAbstractions and no CUDA.jl compatibility layer:
# every numeric operation is a `ccall` into hand-written HIP.
#
# Usage -- the reference API, unchanged, on the reference's own model objects (section "drop-in methods" at the end):
#
#     using AugmentedGaussianProcesses, AGPHip                                              # loading the shim is the switch
#     m = SVGP(kernel, LogisticLikelihood(), AnalyticSVI(1024), Z; optimiser=false)        # the reference constructor
#     m, state = train!(m, X, y, 1000)                  # src/training/training.jl:13-111, now on the MI355X (backend=:cpu opts out)
#     ŷ = predict_y(m, X_test); p = proba_y(m, X_test); μ, σ² = predict_f(m, X_test; cov=true); ELBO(m, X, y)
#     Z = AGPHip.inducingpoints(KmeansAlg(1024), X)      # the device k-means (agp_kmeans) behind InducingPoints' interface
#     om = OnlineSVGP(kernel, GaussianLikelihood(), AnalyticVI(), OIPS(0.9)); train!(om, Xbatch, ybatch; iterations=5)
#
# The explicit form (a wrapped model, for multi-GPU runs and for tests that keep both paths side by side) is still there:
#     hm = AGPHip.HipModel(m); hm, state = train!(hm, X, y, 1000); predict_y(hm, X_test); ELBO(hm, X, y); objective(hm, state, y)
#
# Seams replaced (SURVEY.md section 8b):
#   update_parameters!(model::SVGP / ::MOSVGP, state, x, y)   src/training/training.jl:140-158        -> agp_svgp_cavi_step[_multi]
#   compute_K / compute_κ                                     src/gpblocks/latentgp.jl:205-215         -> inside the step / agp_svgp_refresh_K
#   update_hyperparameters!(m, state, x, y)                   src/hyperparameter/autotuning.jl:86-140  -> agp_svgp_hyper_step[_multi]
#   ELBO(model, state, y) / ELBO(model, X, y)                 src/inference/analyticVI.jl:255-297, src/functions/ELBO.jl:32-47 -> agp_svgp_elbo
#   _predict_f / predict_y / proba_y                          src/training/predictions.jl:25-92,178-247 -> agp_svgp_predict_f / _predict_y / _proba_y
#   init_state                                                src/training/states.jl:1-9               -> agp_svgp_init_state
# Multi-GPU (one Julia process per GPU, e.g. under MPI.jl or Distributed): agp_comm_* binds RCCL inside the library; the host only
# ships the 128-byte id from rank 0 to the others (section "multi-GPU" below).
module AGPHip

using AMDGPU
using AugmentedGaussianProcesses
using KernelFunctions
using LinearAlgebra
using StatsBase: sample, Weights
using Random
using Optimisers
import ProgressMeter                      # (a dependency of the reference: train!'s progress reporting, training.jl:46,71-90)
const AGP = AugmentedGaussianProcesses

import AugmentedGaussianProcesses: train!, predict_f, predict_y, proba_y, ELBO, objective

const libagp = get(ENV, "AGP_HIP_LIB", joinpath(@__DIR__, "..", "augmentedgaussianprocesses.jl_amd", "libagp_hip.so"))

# ---- POD mirrors of include/agp_hip.h (field order and widths are checked by tests/test_host_abi.py on the Python mirror) -------
struct KernelDesc
    kind::Int32
    ard::Int32
    variance::Float64
    scale::Float64
    ard_scales_host::Ptr{Float64}
    has_variance::Int32          # the kernel object is `σ² * k`            (what update_kernel! may step,
    has_transform::Int32         # the kernel object is `k ∘ Scale/ARDTransform`   autotuning_utils.jl:47-67)
end
struct LikDesc
    kind::Int32
    n_class::Int32
    p0::Float64
    p1::Float64
end
struct SvgpDesc
    dtype::Int32
    n_latent::Int32
    latent_offset::Int32
    stochastic::Int32
    m::Int64
    D::Int64
    max_batch::Int64
    lik::LikDesc
    jitter::Float64
    rm_kappa::Float64
    rm_tau::Float64
    elbo_mode::Int32
    flags::Int32                 # AGP_FLAG_STALE_K = 1 : reference_compat_stale_K (training.jl:187-208, SURVEY Appendix A Q1)
end
const AGP_FLAG_STALE_K = Int32(1)
const AGP_SHARD_LATENT, AGP_SHARD_BATCH = Int32(0), Int32(1)

struct AGPError <: Exception
    status::Int32
    msg::String
end
function check(ctx, st::Integer)
    st == 0 && return nothing
    msg = unsafe_string(ccall((:agp_last_error, libagp), Cstring, (Ptr{Cvoid},), ctx))
    if st == 2                                                   # cholesky failure, latentgp.jl:206
        mt = match(r"leading minor (\d+)", msg)
        throw(PosDefException(mt === nothing ? 0 : parse(Int, mt.captures[1])))
    end
    st == 3 && error("K̃ has negative values")                    # latentgp.jl:213
    st == 4 && error(msg)                                        # batch-size check, training.jl:27-29
    st == 6 && throw(ArgumentError(msg))                         # treat_labels!, classification.jl:36-44 / multiclass.jl:81-83
    throw(AGPError(Int32(st), msg))
end

# ---- KernelFunctions.jl objects -> agp_kernel_desc ----------------------------------------------------------------------------
kind(::SqExponentialKernel) = Int32(0)
kind(::Matern52Kernel) = Int32(1)
kind(::Matern32Kernel) = Int32(2)
kind(::ExponentialKernel) = Int32(3)
function kernel_desc(k::Kernel, D::Int)
    σ², hasv = 1.0, Int32(0)
    if k isa ScaledKernel
        σ² = only(k.σ²); k = k.kernel; hasv = Int32(1)
    end
    if k isa TransformedKernel
        t = k.transform; b = k.kernel
        if t isa ScaleTransform
            return KernelDesc(kind(b), 0, σ², only(t.s), C_NULL, hasv, 1), nothing
        elseif t isa ARDTransform
            v = Vector{Float64}(t.v)
            length(v) == D || throw(DimensionMismatch("ARDTransform has $(length(v)) scales, the data $D dimensions"))
            return KernelDesc(kind(b), 1, σ², 1.0, pointer(v), hasv, 1), v   # keep v alive (GC.@preserve at the call)
        end
        error("only ScaleTransform / ARDTransform are wired on the HIP path")
    end
    return KernelDesc(kind(k), 0, σ², 1.0, C_NULL, hasv, 0), nothing
end

# write the (possibly optimised) parameters back into the reference's kernel object (what update_kernel! mutates in place)
function pull_kernel!(k::Kernel, σ²::Float64, scales::Vector{Float64})
    if k isa ScaledKernel
        k.σ² .= σ²; k = k.kernel
    end
    if k isa TransformedKernel
        t = k.transform
        t isa ScaleTransform && (t.s .= scales[1])
        t isa ARDTransform && (t.v .= scales)
    end
    return nothing
end

# opt_noise (gaussian.jl:18-23): p1 carries the ADAM learning rate, σ² is then device state (pull_hypers! mirrors it back)
lik_desc(l::GaussianLikelihood) = LikDesc(0, 1, AGP.noise(l), l.opt_noise === nothing ? 0.0 : Float64(l.opt_noise.eta))
lik_desc(::AGP.BernoulliLikelihood{<:AGP.LogisticLink}) = LikDesc(1, 1, 0.0, 0.0)
lik_desc(l::StudentTLikelihood) = LikDesc(2, 1, l.ν, l.σ)
lik_desc(l::AGP.MultiClassLikelihood{<:AGP.LogisticSoftMaxLink}) = LikDesc(3, AGP.n_class(l), 0.0, 0.0)
lik_desc(l::LaplaceLikelihood) = LikDesc(5, 1, l.β, 0.0)
lik_desc(::AGP.BernoulliLikelihood{<:AGP.SVMLink}) = LikDesc(6, 1, 0.0, 0.0)
lik_desc(l::AGP.PoissonLikelihood{<:AGP.ScaledLogistic}) = LikDesc(7, 1, only(l.invlink.λ), 0.0)   # λ is state: pull_lik_state!
lik_desc(l::NegBinomialLikelihood) = LikDesc(8, 1, Float64(l.r), 0.0)
lik_desc(l::AGP.HeteroscedasticGaussianLikelihood{<:AGP.InvScaledLogistic}) = LikDesc(9, 1, only(l.invlink.λ), 0.0)
lik_desc(l) = error("The $l is not compatible or implemented with AnalyticVI on the HIP path")     # SVGP.jl:48-49

is_multiclass(l) = l isa AGP.MultiClassLikelihood
is_bernoulli(l) = l isa AGP.BernoulliLikelihood
is_event(l) = l isa AGP.PoissonLikelihood || l isa NegBinomialLikelihood

# ---- the device twin of a reference model -------------------------------------------------------------------------------------
mutable struct HipModel{T,M<:AGP.AbstractGPModel{T}}
    model::M                       # the reference object: kernels, Z, likelihood, inference, posterior live on in it
    ctx::Ptr{Cvoid}
    h::Ptr{Cvoid}
    comm::Ptr{Cvoid}               # agp_comm* (C_NULL: single GPU)
    shard::Int32                   # AGP_SHARD_LATENT / AGP_SHARD_BATCH when comm is set
    latent_range::UnitRange{Int}   # latents of model.f held by this rank (all of them unless latent-parallel)
    maxbatch::Int
    X::Union{Nothing,ROCMatrix{T}} # D x N (ColVecs memory order == the ABI's point-major layout)
    y::Any                         # ROCVector{T} / ROCVector{Int32} (class index) / ROCMatrix{T} (n_task x N, multi-output)
    N::Int
    last_idx::Any                  # device indices of the last minibatch (kept alive: the step is asynchronous)
    stale_K::Bool
    rank::Int32                    # place in a multi-GPU run (comm_init!): re-applied whenever the handle is re-created
    world::Int32
    keep::Vector{Any}              # index buffers a queued look-ahead / pending step may still read (two generations)
end

is_mo(hm::HipModel) = hm.model isa AGP.MOSVGP
nlat(hm::HipModel) = length(hm.latent_range)

"""
    HipModel(model::Union{SVGP,MOSVGP}; device=AMDGPU.device_id()-1, reference_compat_stale_K=false,
             latent_range=1:length(model.f))

Wrap a reference model.  Nothing is allocated on the device until data arrive (`train!`, `ELBO`, `predict_*`).
"""
function HipModel(model::M; reference_compat_stale_K::Bool=false,
                  latent_range::UnitRange{Int}=1:length(model.f)) where {T,M<:AGP.AbstractGPModel{T}}
    model isa Union{SVGP,AGP.MOSVGP} || error("only SVGP / MOSVGP run on the HIP path")
    AGP.inference(model) isa AnalyticVI || error("The inference object should be of type `AnalyticVI`")   # SVGP.jl:45-47
    return HipModel{T,M}(model, C_NULL, C_NULL, C_NULL, AGP_SHARD_LATENT, latent_range, 0, nothing, nothing, 0, nothing,
                         reference_compat_stale_K, Int32(0), Int32(1), Any[])
end

function ensure_ctx!(hm::HipModel)
    hm.ctx == C_NULL || return hm.ctx
    ctx = Ref{Ptr{Cvoid}}()
    stream = AMDGPU.stream().stream                          # share AMDGPU.jl's HIP stream with the library
    st = ccall((:agp_ctx_create, libagp), Int32, (Int32, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), AMDGPU.device_id() - 1, stream, ctx)
    st == 0 || throw(AGPError(st, "agp_ctx_create"))
    hm.ctx = ctx[]
    finalizer(destroy!, hm)
    return hm.ctx
end

function destroy!(hm::HipModel)
    hm.h == C_NULL || ccall((:agp_svgp_destroy, libagp), Int32, (Ptr{Cvoid},), hm.h)
    hm.comm == C_NULL || ccall((:agp_comm_destroy, libagp), Int32, (Ptr{Cvoid},), hm.comm)
    hm.ctx == C_NULL || ccall((:agp_ctx_destroy, libagp), Int32, (Ptr{Cvoid},), hm.ctx)
    hm.h = hm.comm = hm.ctx = C_NULL
    return nothing
end

# (re)create the device handle for batches up to `maxbatch`; the posterior and the optimiser counters travel (get/set_state)
function ensure_handle!(hm::HipModel{T}, maxbatch::Int) where {T}
    (hm.h != C_NULL && maxbatch <= hm.maxbatch) && return hm.h
    ctx = ensure_ctx!(hm)
    model = hm.model
    old = hm.h == C_NULL ? nothing : (pull_posterior!(hm); opt_state(hm))
    hm.h == C_NULL || ccall((:agp_svgp_destroy, libagp), Int32, (Ptr{Cvoid},), hm.h)
    inf = AGP.inference(model)
    gp1 = model.f[first(hm.latent_range)]
    D = length(first(AGP.Zview(gp1)))
    m = AGP.dim(gp1)
    stoch = AGP.is_stochastic(inf)
    rm = stoch ? inf.vi_opt.optimiser : nothing             # RobbinsMonro(κ, τ), optimisers.jl:1-19
    stoch && !(rm isa AGP.RobbinsMonro) && error("only RobbinsMonro is wired on this path (ALRSVI is dead code in the reference)")
    ld = is_mo(hm) ? LikDesc(4, 1, 0.0, 0.0) : lik_desc(AGP.likelihood(model))
    desc = SvgpDesc(T == Float64 ? 0 : 1, nlat(hm), first(hm.latent_range) - 1, stoch ? 1 : 0, m, D, maxbatch, ld, 0.0,
                    stoch ? rm.κ : 0.51, stoch ? rm.τ : 1.0, 0, hm.stale_K ? AGP_FLAG_STALE_K : Int32(0))
    h = Ref{Ptr{Cvoid}}()
    check(ctx, ccall((:agp_svgp_create, libagp), Int32, (Ptr{Cvoid}, Ref{SvgpDesc}, Ptr{Ptr{Cvoid}}), ctx, desc, h))
    hm.h, hm.maxbatch = h[], maxbatch
    for (i, q) in enumerate(hm.latent_range)
        gp = model.f[q]
        kd, keep = kernel_desc(AGP.kernel(gp), D)
        GC.@preserve keep check(ctx, ccall((:agp_svgp_set_kernel, libagp), Int32, (Ptr{Cvoid}, Int32, Ref{KernelDesc}), hm.h, i - 1, kd))
        Zd = ROCArray{T}(reduce(hcat, AGP.Zview(gp)))       # D x m, point-major
        check(ctx, ccall((:agp_svgp_set_Z, libagp), Int32, (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Int64), hm.h, i - 1, pointer(Zd), D))
        μ₀ = AGP.pr_mean(gp)
        if !(μ₀ isa AGP.ZeroMean)   # (with an optimiser the FIRST HYPER STEP errors, like the reference's own broken update: update_hyperparameters!)
            v = ROCArray{T}(μ₀(AGP.Zview(gp)))
            check(ctx, ccall((:agp_svgp_set_prior_mean, libagp), Int32, (Ptr{Cvoid}, Int32, Ptr{Cvoid}), hm.h, i - 1, pointer(v)))
        end
        AMDGPU.synchronize()
    end
    if is_mo(hm)
        liks = collect(AGP.likelihood(model))
        lds = [lik_desc(l) for l in liks]
        Q = length(model.f)
        A = Matrix{Float64}(undef, length(liks), Q)          # model.A[t][j][q]; nf_per_task == 1 on this path
        for t in 1:length(liks), q in 1:Q
            A[t, q] = model.A[t][1][q]
        end
        Aopt = model.A_opt
        nlat(hm) == Q || check(ctx, ccall((:agp_svgp_mo_shard, libagp), Int32, (Ptr{Cvoid}, Int32), hm.h, Q))
        check(ctx, ccall((:agp_svgp_set_multioutput, libagp), Int32,
                         (Ptr{Cvoid}, Int32, Ptr{LikDesc}, Ptr{Float64}, Float64, Float64, Float64, Float64),
                         hm.h, length(liks), lds, permutedims(A), Aopt === nothing ? 0.0 : Aopt.eta, 0.9, 0.999, 1e-8))
    elseif AGP.likelihood(model) isa AGP.PoissonLikelihood    # the λ update integrates logistic by Gauss-Hermite (utils.jl:16-19)
        check(ctx, ccall((:agp_svgp_set_quadrature, libagp), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32),
                         hm.h, AGP.pred_nodes, AGP.pred_weights, length(AGP.pred_nodes)))
    end
    # hyper-parameter optimisers: SVGP(...; optimiser, Zoptimiser) (SVGP.jl:39-42, default ADAM(0.01) / nothing)
    ko, zo = AGP.opt(gp1), AGP.Zopt(gp1)
    if ko !== nothing || zo !== nothing
        # the reference hands whatever Optimisers.jl rule it is given to Optimisers.apply (autotuning_utils.jl:47-82); the device
        # carries ADAM, Descent and Momentum (agp_svgp_hyper_rule), anything else is refused here rather than silently replaced
        rule(o) = o === nothing || o isa Optimisers.ADAM ? (0, 0.0) : o isa Optimisers.Descent ? (1, 0.0) :
                  o isa Optimisers.Momentum ? (2, Float64(o.rho)) :
                  error("hyper-parameter optimiser $(typeof(o)) is not available on the device (ADAM, Descent, Momentum are)")
        (kr, kρ), (zr, zρ) = rule(ko), rule(zo)
        adam = ko isa Optimisers.ADAM ? ko : zo isa Optimisers.ADAM ? zo : Optimisers.ADAM()
        check(ctx, ccall((:agp_svgp_hyper_configure, libagp), Int32,
                         (Ptr{Cvoid}, Int32, Float64, Int32, Float64, Float64, Float64, Float64),
                         hm.h, ko === nothing ? 0 : 1, ko === nothing ? 0.0 : ko.eta, zo === nothing ? 0 : 1,
                         zo === nothing ? 0.0 : zo.eta, adam.beta[1], adam.beta[2], 1e-8))
        check(ctx, ccall((:agp_svgp_hyper_rule, libagp), Int32, (Ptr{Cvoid}, Int32, Float64, Int32, Float64), hm.h, kr, kρ, zr, zρ))
    end
    if old !== nothing
        push_posterior!(hm)
        check(ctx, ccall((:agp_svgp_set_opt_state, libagp), Int32, (Ptr{Cvoid}, Int64), hm.h, old))
    end
    # a handle created (or re-created for a larger batch) AFTER comm_init! must know its shard too: the once-per-evaluation ELBO
    # terms and the Gaussian-KL weight of the hyper-gradient depend on it (the *_multi calls also take it from the communicator)
    if hm.comm != C_NULL && hm.shard == AGP_SHARD_BATCH
        check(ctx, ccall((:agp_svgp_set_batch_shard, libagp), Int32, (Ptr{Cvoid}, Int32, Int32), hm.h, hm.rank, hm.world))
    end
    return hm.h
end

opt_state(hm::HipModel) = (n = Ref{Int64}(); ccall((:agp_svgp_get_opt_state, libagp), Int32, (Ptr{Cvoid}, Ref{Int64}), hm.h, n); n[])

# (μ, Σ, η₁, η₂) device -> the reference's VarPosterior (posterior.jl:21-27), e.g. at the end of train!
function pull_posterior!(hm::HipModel{T}) where {T}
    for (i, q) in enumerate(hm.latent_range)
        gp = hm.model.f[q]
        m = AGP.dim(gp)
        μ = ROCVector{T}(undef, m); η₁ = ROCVector{T}(undef, m)
        Σ = ROCMatrix{T}(undef, m, m); η₂ = ROCMatrix{T}(undef, m, m)
        check(hm.ctx, ccall((:agp_svgp_get_state, libagp), Int32, (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                            hm.h, i - 1, pointer(μ), pointer(Σ), pointer(η₁), pointer(η₂)))
        ccall((:agp_ctx_sync, libagp), Int32, (Ptr{Cvoid},), hm.ctx)
        gp.post.μ .= Array(μ); gp.post.η₁ .= Array(η₁)
        gp.post.Σ.data .= Array(Σ); gp.post.η₂.data .= Array(η₂)   # symmetric: row-major == column-major
    end
    return hm
end
# the reference's (η₁, η₂) -> device (a model trained on the CPU continues on the GPU)
function push_posterior!(hm::HipModel{T}) where {T}
    for (i, q) in enumerate(hm.latent_range)
        gp = hm.model.f[q]
        η₁ = ROCArray{T}(AGP.nat1(gp)); η₂ = ROCArray{T}(Matrix(AGP.nat2(gp)))
        check(hm.ctx, ccall((:agp_svgp_set_state, libagp), Int32, (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Cvoid}), hm.h, i - 1, pointer(η₁), pointer(η₂)))
    end
    return hm
end

# kernel parameters / Z (device, after hyper steps) -> the reference objects; λ of Poisson / Heteroscedastic likewise
function pull_hypers!(hm::HipModel{T}) where {T}
    for (i, q) in enumerate(hm.latent_range)
        gp = hm.model.f[q]
        (AGP.opt(gp) === nothing && AGP.Zopt(gp) === nothing) && continue
        D = length(first(AGP.Zview(gp))); m = AGP.dim(gp)
        σ² = Ref{Float64}(); sc = Vector{Float64}(undef, D)
        check(hm.ctx, ccall((:agp_svgp_get_kernel, libagp), Int32, (Ptr{Cvoid}, Int32, Ref{Float64}, Ptr{Float64}), hm.h, i - 1, σ², sc))
        pull_kernel!(AGP.kernel(gp), σ²[], sc)
        Zd = ROCMatrix{T}(undef, D, m)
        check(hm.ctx, ccall((:agp_svgp_get_Z, libagp), Int32, (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Int64), hm.h, i - 1, pointer(Zd), D))
        Zh = Array(Zd)
        for j in 1:m
            AGP.Zview(gp)[j] .= view(Zh, :, j)
        end
    end
    l = AGP.likelihood(hm.model)
    if l isa Union{AGP.PoissonLikelihood,AGP.HeteroscedasticGaussianLikelihood}
        v = Ref{Float64}()
        check(hm.ctx, ccall((:agp_svgp_get_lik_param, libagp), Int32, (Ptr{Cvoid}, Ref{Float64}), hm.h, v))
        l.invlink.λ .= v[]                                   # poisson.jl:78, heteroscedastic.jl:95 mutate it in place
    end
    if l isa GaussianLikelihood && l.opt_noise !== nothing   # gaussian.jl:69 mutates l.σ² in place
        v = Ref{Float64}()
        check(hm.ctx, ccall((:agp_svgp_get_lik_param, libagp), Int32, (Ptr{Cvoid}, Ref{Float64}), hm.h, v))
        l.σ² .= v[]
    end
    if is_mo(hm)
        liks = AGP.likelihood(hm.model); Q = length(hm.model.f)
        A = Matrix{Float64}(undef, Q, length(liks))          # row-major n_task x Q on the C side
        check(hm.ctx, ccall((:agp_svgp_get_A, libagp), Int32, (Ptr{Cvoid}, Ptr{Float64}), hm.h, A))
        for t in 1:length(liks), q in 1:Q
            hm.model.A[t][1][q] = A[q, t]
        end
    end
    return hm
end

# ---- data ---------------------------------------------------------------------------------------------------------------------
# wrap_X / wrap_data (src/data/datacontainer.jl:18-74, src/data/utils.jl:16-30): labels are normalised by the reference's own
# treat_labels!, then laid out for the device: ±1 / real T[N]; class index Int32[N] (0-based row-wise findfirst of the one-hot
# BitMatrix, multiclass.jl:81-94); multi-output T[n_task x N] (point-major)
function upload!(hm::HipModel{T}, X::AbstractMatrix, y; obsdim::Int=1) where {T}
    Xv, _ = AGP.wrap_X(X, obsdim)
    data = AGP.wrap_data(Xv, y, AGP.likelihood(hm.model))     # runs treat_labels! (ArgumentErrors surface here)
    yt = AGP.output(data)
    hm.X = ROCArray{T}(obsdim == 1 ? permutedims(X) : X)      # one-time RowVecs -> point-major permutation
    hm.N = size(hm.X, 2)
    if is_mo(hm)
        hm.y = ROCArray{T}(permutedims(reduce(hcat, yt)))     # n_task x N
    elseif is_multiclass(AGP.likelihood(hm.model))
        hm.y = ROCArray(Int32.(map(r -> findfirst(r) - 1, eachrow(yt))))
    else
        hm.y = ROCArray{T}(yt)
    end
    return data
end

# ---- update_parameters!(model, state, x, y) (training.jl:140-158) ---------------------------------------------------------------
# `idx` is the minibatch drawn by train! (StatsBase.sample, training.jl:51-53), 1-based; nothing = full batch.
# minibatch indices on the device, 0-based (an already uploaded vector is taken as it is: the look-ahead and the step that
# consumes it must see the same buffer)
device_indices(idx::Nothing) = nothing
device_indices(idx::ROCArray{Int64}) = idx
device_indices(idx::AbstractVector{<:Integer}) = ROCArray(Int64.(idx .- 1))

function update_parameters!(hm::HipModel{T}, idx, ρ::Real) where {T}
    D = size(hm.X, 1)
    idd = device_indices(idx)
    hm.last_idx = idd
    B = idx === nothing ? hm.N : length(idx)
    idp = idd === nothing ? Ptr{Int64}(C_NULL) : pointer(idd)
    st = if hm.comm == C_NULL && nlat(hm) == length(hm.model.f)
        ccall((:agp_svgp_cavi_step, libagp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Int64}, Int64, Float64),
              hm.h, pointer(hm.X), D, pointer(hm.y), idp, B, ρ)
    else
        ccall((:agp_svgp_cavi_step_multi, libagp), Int32,
              (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Int64}, Int64, Float64),
              hm.h, hm.comm, hm.shard, pointer(hm.X), D, pointer(hm.y), idp, B, ρ)
    end
    check(hm.ctx, st)
    return nothing
end

# Look-ahead: Knm / kappa of the NEXT minibatch on the library's second stream, next to the current step's factorisation
# (agp_svgp_prefetch; a no-op while K is stale, i.e. right after a hyper-parameter step).  `idd` must be the very buffer the next
# update_parameters! passes.
function prefetch!(hm::HipModel, idd::ROCArray{Int64})
    D = size(hm.X, 1)
    check(hm.ctx, ccall((:agp_svgp_prefetch, libagp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Int64}, Int64),
                        hm.h, pointer(hm.X), D, pointer(idd), length(idd)))
    return nothing
end

# update_hyperparameters!(m, state, x, y) (autotuning.jl:86-140) on the minibatch of the last step
function update_hyperparameters!(hm::HipModel; tied::Bool=false)
    any(gp -> !(AGP.pr_mean(gp) isa AGP.ZeroMean), hm.model.f[hm.latent_range]) &&
        error("a non-zero prior mean with hyper-parameter optimisation is not wired: the reference's own prior-mean update " *
              "(autotuning.jl:104-106 vs src/mean/constantmean.jl:31) cannot run")
    st = hm.comm == C_NULL && !tied ? ccall((:agp_svgp_hyper_step, libagp), Int32, (Ptr{Cvoid},), hm.h) :
         ccall((:agp_svgp_hyper_step_multi, libagp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32), hm.h, hm.comm, tied ? 1 : 0)
    check(hm.ctx, st)
end

"""
    hyper_counters(hm) -> (gradients, gradients_with_one_product_G_K)

Diagnostics of the hyper-parameter iteration (`agp_svgp_hyper_counters`, round 4): how many hyper-gradients this handle has
evaluated, and how many of them formed `G_K` from one m^3 product (DESIGN.md section 6) instead of `kappa' H` and `K^-1 Sigma K^-1`.
"""
function hyper_counters(hm::HipModel)
    ng = Ref{Int64}(0); nf = Ref{Int64}(0)
    check(hm.ctx, ccall((:agp_svgp_hyper_counters, libagp), Int32, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), hm.h, ng, nf))
    return ng[], nf[]
end

"""
    train!(hm::HipModel, X, y, iterations=100; callback=nothing, state=nothing, obsdim=1, idx_stream=nothing)

`train!` of the reference (src/training/training.jl:13-111) with the step on the device: same checks, same ρ = N/B, same
`StatsBase.sample(1:N, B; replace=false)` per iteration (or the caller's `idx_stream`, one index vector per iteration, so that
runs are reproducible across back-ends), same hyper-step gating (`n_iter % atfrequency == 0 && n_iter >= 3 &&
local_iter != iterations`, :65-69), fixed iteration count (the reference never reads `convergence` / ϵ, :48,93-94), final
`compute_Ks` (:107).  Returns `(hm, state)`; the state is the device-resident one: pass it back (`state=state`) to continue, or
omit it to restart counters and local variables like the reference's `init_state` (states.jl:1-9).
"""
function train!(hm::HipModel{T}, X::AbstractArray, y, iterations::Int=100; callback=nothing, convergence=nothing,
                state=nothing, obsdim=1, idx_stream=nothing) where {T}
    iterations > 0 || error("Number of iterations should be positive")
    model = hm.model
    inf = AGP.inference(model)
    data = upload!(hm, X isa AbstractMatrix ? X : reduce(hcat, X)', y; obsdim)
    N = hm.N
    if AGP.is_stochastic(model)
        0 < AGP.batchsize(inf) <= N || error(
            "The size of mini-batch $(AGP.batchsize(inf)) is incorrect (negative or bigger than number of samples), please set `batchsize` correctly in the inference object")
        AGP.set_ρ!(model, N / AGP.batchsize(inf))
    else
        AGP.set_batchsize!(inf, N)
    end
    B = AGP.batchsize(inf)
    Blocal = (hm.comm != C_NULL && hm.shard == AGP_SHARD_BATCH) ? B ÷ comm_world(hm) : B
    ensure_handle!(hm, Blocal)
    if state === nothing
        AGP.setHPupdated!(inf, true)
        check(hm.ctx, ccall((:agp_svgp_init_state, libagp), Int32, (Ptr{Cvoid},), hm.h))      # init_state(model), training.jl:41-45
    else
        check(hm.ctx, ccall((:agp_svgp_invalidate_data, libagp), Int32, (Ptr{Cvoid},), hm.h)) # the buffers were re-uploaded
    end
    check(hm.ctx, ccall((:agp_svgp_refresh_K, libagp), Int32, (Ptr{Cvoid},), hm.h))
    ρ = AGP.is_stochastic(model) ? N / B : 1.0
    hyper_on = any(gp -> AGP.opt(gp) !== nothing || AGP.Zopt(gp) !== nothing, model.f)
    # minibatch of iteration `it` on the device: the reference's sample(1:N, B; replace=false) (training.jl:51-53), drawn in the
    # same order, one iteration ahead of its use so that the look-ahead can run next to the current step
    function draw(it)
        AGP.is_stochastic(model) || return nothing
        idx = idx_stream === nothing ? sample(1:N, B; replace=false) : idx_stream[it]
        if hm.comm != C_NULL && hm.shard == AGP_SHARD_BATCH    # this rank's share of the minibatch (rho stays N / B_total)
            r = comm_rank(hm)
            idx = idx[(r * Blocal + 1):((r + 1) * Blocal)]
        end
        return device_indices(idx)
    end
    if AGP.verbose(model) > 0   # training.jl:35-38
        @info "Starting training $model with $N samples, $(size(hm.Xd, 1)) features and $(length(model.f)) latent GP" *
              (length(model.f) > 1 ? "s" : "")
    end
    local_iter = 1
    # progress reporting of the reference (training.jl:46,71-90): ProgressMeter with (:iter, :ELBO); the ELBO is
    # `objective(model, state, y)` on the kernel matrices / local variables of the step just taken -- verbose == 2: every 10th
    # iteration, verbose > 2: every iteration.  On the device the value is ENQUEUED behind the step (agp_svgp_elbo_enqueue) and shown
    # one report later, so that reporting never makes the host wait for the stream inside the loop.
    prog = AGP.verbose(model) > 1 ? ProgressMeter.Progress(iterations; dt=0.2, desc="Training Progress: ") : nothing
    pending_elbo = nothing   # (ticket, iteration it belongs to)
    function report(it)
        prog === nothing && return
        (AGP.verbose(model) > 2 || it % 10 == 0) || return
        tk = objective_enqueue(hm)
        if pending_elbo !== nothing
            elbo = objective_fetch(hm, pending_elbo[1])
            AGP.verbose(model) == 2 && ProgressMeter.update!(prog, pending_elbo[2] - 1)
            ProgressMeter.next!(prog; showvalues=[(:iter, pending_elbo[2]), (:ELBO, elbo)])
        end
        pending_elbo = (tk, it)
    end
    idd = draw(1)
    while true
        stepped = false   # this iteration's variational update has been enqueued (the device's own counters have moved on)
        try
        update_parameters!(hm, idd, ρ)
        stepped = true
        AGP.set_trained!(model, true)
        # the next minibatch is drawn now and its look-ahead enqueued right behind the step (next to its factorisation) -- unless a
        # hyper step follows: it moves the kernel / Z, a look-ahead against the old ones would be thrown away (agp_svgp_prefetch
        # contract, include/agp_hip.h; the Python mirror skips it the same way, svgp.py train_)
        hyper_now = hyper_on && (AGP.n_iter(model) % model.atfrequency == 0) && (AGP.n_iter(model) >= 3) && (local_iter != iterations)
        nxt = local_iter < iterations ? draw(local_iter + 1) : nothing
        # index buffers stay referenced for two iterations: the look-ahead reads `nxt`, and the step's natural-gradient part is
        # taken by the NEXT launch (it reads device copies only, but the buffer identity is what the look-ahead is recognised by)
        push!(hm.keep, idd); length(hm.keep) > 3 && popfirst!(hm.keep)
        (nxt === nothing || hyper_now) || prefetch!(hm, nxt)
        callback === nothing || callback(hm, hm, AGP.n_iter(model))
        hyper_now && update_hyperparameters!(hm)
        report(local_iter)
        local_iter += 1
        inf.n_iter += 1
        (local_iter <= iterations) || break
        idd = nxt
        catch e
            # training.jl:95-101: an InterruptException ends the loop with a warning, everything else is rethrown.  A variational
            # update that was already enqueued counts (its Robbins-Monro step has been taken on the device); the pending natural-
            # gradient step is taken by agp_svgp_check_status below, and `state=` continues from there.
            if isa(e, InterruptException)
                @warn "Training interrupted by user at iteration $local_iter"
                if stepped
                    local_iter += 1
                    inf.n_iter += 1
                end
                break
            else
                rethrow(e)
            end
        end
    end
    pending_elbo === nothing || objective_fetch(hm, pending_elbo[1])   # (close the last report's ticket)
    if AGP.verbose(model) > 0   # training.jl:103-105
        @info "Training ended after $(local_iter - 1) iterations. Total number of iterations $(AGP.n_iter(model))"
    end
    check(hm.ctx, ccall((:agp_svgp_check_status, libagp), Int32, (Ptr{Cvoid},), hm.h))
    check(hm.ctx, ccall((:agp_svgp_refresh_K, libagp), Int32, (Ptr{Cvoid},), hm.h))            # compute_Ks, training.jl:107
    pull_posterior!(hm)                                       # the reference object is usable on the CPU again
    pull_hypers!(hm)
    AGP.set_trained!(model, true)
    return hm, hm
end

# ---- ELBO ------------------------------------------------------------------------------------------------------------------------
# objective(model, state, y) = ELBO(model, state, y) on the last minibatch (SVGP.jl:90, analyticVI.jl:255-297)
function objective(hm::HipModel, state=nothing, y=nothing)
    out = Ref{Float64}()
    if hm.comm != C_NULL
        check(hm.ctx, ccall((:agp_svgp_elbo_multi, libagp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ref{Float64}), hm.h, hm.comm, hm.shard, out))
        return out[]
    end
    B = hm.last_idx === nothing ? hm.N : length(hm.last_idx)
    ρ = AGP.is_stochastic(hm.model) ? hm.N / B : 1.0
    idp = hm.last_idx === nothing ? Ptr{Int64}(C_NULL) : pointer(hm.last_idx)
    check(hm.ctx, ccall((:agp_svgp_elbo, libagp), Int32,
                        (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Int64}, Int64, Float64, Int32, Ref{Float64}),
                        hm.h, pointer(hm.X), size(hm.X, 1), pointer(hm.y), idp, B, ρ, 0, out))
    return out[]
end

# the same evaluation put into the stream without waiting for it (agp_svgp_elbo_enqueue / agp_svgp_elbo_fetch): a callback that
# monitors convergence enqueues `t = objective_enqueue(hm)` and reads `objective_fetch(hm, t)` an iteration or ten later, so the
# training loop never synchronises with the device (up to 8 tickets in flight)
function objective_enqueue(hm::HipModel)
    B = hm.last_idx === nothing ? hm.N : length(hm.last_idx)
    ρ = AGP.is_stochastic(hm.model) ? hm.N / B : 1.0
    idp = hm.last_idx === nothing ? Ptr{Int64}(C_NULL) : pointer(hm.last_idx)
    t = Ref{Int32}()
    check(hm.ctx, ccall((:agp_svgp_elbo_enqueue, libagp), Int32,
                        (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Int64}, Int64, Float64, Int32, Ref{Int32}),
                        hm.h, pointer(hm.X), size(hm.X, 1), pointer(hm.y), idp, B, ρ, 0, t))
    return t[]
end
function objective_fetch(hm::HipModel, ticket::Integer; wait::Bool=true)
    out, ready = Ref{Float64}(), Ref{Int32}()
    check(hm.ctx, ccall((:agp_svgp_elbo_fetch, libagp), Int32, (Ptr{Cvoid}, Int32, Int32, Ref{Float64}, Ref{Int32}),
                        hm.h, ticket, wait ? 1 : 0, out, ready))
    return ready[] == 1 ? out[] : nothing
end

"""
    task_graph_fallbacks(hm) -> Int

How many factorisation launches of this model's context lost a tile dependency and were re-run by their in-stream fallback since the
context was created (`agp_ctx_task_graph_fallbacks`, round 6): 0 on a GPU the process has to itself.  Synchronises the stream.
"""
function task_graph_fallbacks(hm::HipModel)
    n = Ref{Int64}(0)
    check(hm.ctx, ccall((:agp_ctx_task_graph_fallbacks, libagp), Int32, (Ptr{Cvoid}, Ref{Int64}), hm.ctx, n))
    return Int(n[])
end

"""
    SideObjective(hm::HipModel, max_eval_batch; ring=4)
    ticket = enqueue!(side, idx::ROCVector{Int64}, ρ)     # snapshot of (η₁, η₂) now, ELBO(model, X[idx], y[idx]) on the side stream
    value  = fetch(side, ticket; wait=true)

Convergence monitoring NEXT TO the training stream (round 6; the Python mirror's `agp_amd.SideObjective`, same three ABI calls): a
shadow `HipModel` of a deep copy of the reference model on a HIP stream and context of its own, its K_ZZ factored once; `enqueue!`
copies (η₁, η₂) out of the training handle on the training stream (`agp_svgp_get_state`, two device copies), makes the side stream wait
for them (`AMDGPU.HIP` event), installs them there (`agp_svgp_set_state`) and enqueues the evaluation (`agp_svgp_elbo_enqueue`, fresh
local variables: `ELBO(model, X, y)` of src/functions/ELBO.jl:32-47 with ρ explicit).  Replaces the in-line `objective(model, state, y)`
of `train!`'s progress reporting (src/training/training.jl:71-90) where the host must not wait and the training stream must not carry
the evaluation.  Kernels and inducing points must be fixed (`optimiser=false, Zoptimiser=false`): the shadow's K_ZZ is factored once.
The values equal the in-line evaluation's to a few ulp; the training trajectory with snapshots equals the one without to rounding.
"""
mutable struct SideObjective{T}
    hm::HipModel{T}
    shadow::HipModel{T}
    stream::AMDGPU.HIPStream
    ring::Vector{Tuple{ROCVector{T},ROCMatrix{T},AMDGPU.HIP.HIPEvent,AMDGPU.HIP.HIPEvent}}
    n::Int
end

function SideObjective(hm::HipModel{T}, max_eval_batch::Int; ring::Int=4) where {T}
    (length(hm.model.f) == 1 && !is_mo(hm)) || error("SideObjective: single-latent SVGP models")
    all(gp -> AGP.opt(gp) === nothing && AGP.Zopt(gp) === nothing, hm.model.f) ||
        error("SideObjective: kernels and inducing points must be fixed (optimiser=false, Zoptimiser=false)")
    hm.h != C_NULL || error("SideObjective: the model has no device state yet")
    stream = AMDGPU.HIPStream()
    shadow = HipModel(deepcopy(hm.model))
    AMDGPU.stream!(stream) do                                  # the shadow's context takes AMDGPU.jl's current stream: the side stream
        shadow.X, shadow.y, shadow.N = hm.X, hm.y, hm.N        # (the same device data: read-only on both sides)
        ensure_handle!(shadow, max_eval_batch)
        check(shadow.ctx, ccall((:agp_svgp_refresh_K, libagp), Int32, (Ptr{Cvoid},), shadow.h))
    end
    AMDGPU.synchronize(stream)
    m = AGP.dim(hm.model.f[1])
    slots = [(ROCVector{T}(undef, m), ROCMatrix{T}(undef, m, m), AMDGPU.HIP.HIPEvent(stream), AMDGPU.HIP.HIPEvent(stream)) for _ in 1:ring]
    return SideObjective{T}(hm, shadow, stream, slots, 0)
end

function enqueue!(side::SideObjective{T}, idx::ROCVector{Int64}, ρ::Real) where {T}
    hm, sh = side.hm, side.shadow
    e1, e2, ev_snap, ev_used = side.ring[side.n % length(side.ring) + 1]
    main = AMDGPU.stream()
    side.n >= length(side.ring) && AMDGPU.HIP.wait(ev_used, main)   # the side stream has installed what this slot held
    check(hm.ctx, ccall((:agp_svgp_get_state, libagp), Int32, (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                        hm.h, 0, C_NULL, C_NULL, pointer(e1), pointer(e2)))
    AMDGPU.HIP.record(ev_snap, main)
    tk = Ref{Int32}()
    AMDGPU.stream!(side.stream) do
        AMDGPU.HIP.wait(ev_snap, side.stream)
        check(sh.ctx, ccall((:agp_svgp_set_state, libagp), Int32, (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Cvoid}), sh.h, 0, pointer(e1), pointer(e2)))
        AMDGPU.HIP.record(ev_used, side.stream)
        check(sh.ctx, ccall((:agp_svgp_elbo_enqueue, libagp), Int32,
                            (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Int64}, Int64, Float64, Int32, Ref{Int32}),
                            sh.h, pointer(hm.X), size(hm.X, 1), pointer(hm.y), pointer(idx), length(idx), Float64(ρ), 1, tk))
    end
    side.n += 1
    return tk[]
end

Base.fetch(side::SideObjective, ticket::Integer; wait::Bool=true) = objective_fetch(side.shadow, ticket; wait)

# external ELBO(model, X, y) (src/functions/ELBO.jl:28-47): kernel matrices recomputed on (X, y), fresh local variables, one local
# update.  The reference keeps ρ = N/B of the last train! here (Appendix A Q13); pass ρ = 1 for the properly scaled value.
function ELBO(hm::HipModel{T}, X::AbstractMatrix, y::AbstractArray; obsdim=1, ρ::Real=AGP.ρ(AGP.inference(hm.model))) where {T}
    twin = HipModel{T,typeof(hm.model)}(hm.model, hm.ctx, hm.h, C_NULL, hm.shard, hm.latent_range, hm.maxbatch, nothing, nothing,
                                        0, nothing, hm.stale_K, hm.rank, hm.world, Any[])  # same handle, its own data buffers
    upload!(twin, X, y; obsdim)
    n = twin.N
    if n > hm.maxbatch
        ensure_handle!(hm, n); twin.h = hm.h
    end
    out = Ref{Float64}()
    check(hm.ctx, ccall((:agp_svgp_elbo, libagp), Int32,
                        (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Int64}, Int64, Float64, Int32, Ref{Float64}),
                        hm.h, pointer(twin.X), size(twin.X, 1), pointer(twin.y), C_NULL, n, ρ, 1, out))
    return out[]
end

# ---- prediction (src/training/predictions.jl) --------------------------------------------------------------------------------------
function _predict_f_dev(hm::HipModel{T}, Xt::AbstractMatrix; cov::Bool, obsdim::Int=1) where {T}
    ensure_handle!(hm, max(hm.maxbatch, 1))
    Xd = ROCArray{T}(obsdim == 1 ? permutedims(Xt) : Xt)
    D, nt = size(Xd)
    nout = is_mo(hm) ? length(AGP.likelihood(hm.model)) : nlat(hm)
    μ = ROCMatrix{T}(undef, nt, nout)                        # column l == latent / task l  (ABI: T[n_out][n_t])
    v = cov ? ROCMatrix{T}(undef, nt, nout) : nothing
    check(hm.ctx, ccall((:agp_svgp_predict_f, libagp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                        hm.h, pointer(Xd), D, nt, pointer(μ), cov ? pointer(v) : C_NULL))
    ccall((:agp_ctx_sync, libagp), Int32, (Ptr{Cvoid},), hm.ctx)
    return Array(μ), (cov ? Array(v) : nothing), Xd
end

# predict_f(model, X_test; cov=false, diag=true) predictions.jl:141-164: Vector (one latent) or Tuple of Vectors
function predict_f(hm::HipModel{T}, X_test::AbstractMatrix, state=nothing; cov::Bool=false, diag::Bool=true, obsdim::Int=1) where {T}
    if cov && !diag                                           # full covariance, predictions.jl:45-49 (K*m materialised: small n_t)
        Xd = ROCArray{T}(obsdim == 1 ? permutedims(X_test) : X_test)
        D, nt = size(Xd); L = nlat(hm)
        μ = ROCMatrix{T}(undef, nt, L); Σ = ROCArray{T}(undef, nt, nt, L)
        check(hm.ctx, ccall((:agp_svgp_predict_f_cov, libagp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                            ensure_handle!(hm, max(hm.maxbatch, 1)), pointer(Xd), D, nt, pointer(μ), pointer(Σ)))
        ccall((:agp_ctx_sync, libagp), Int32, (Ptr{Cvoid},), hm.ctx)
        μh, Σh = Array(μ), Array(Σ)
        return L == 1 ? (μh[:, 1], Symmetric(Σh[:, :, 1])) : (Tuple(μh[:, l] for l in 1:L), Tuple(Symmetric(Σh[:, :, l]) for l in 1:L))
    end
    μ, v, _ = _predict_f_dev(hm, X_test; cov, obsdim)
    n = size(μ, 2)
    if n == 1 && !is_mo(hm)
        return cov ? (μ[:, 1], v[:, 1]) : μ[:, 1]
    end
    μt = Tuple(μ[:, l] for l in 1:n)
    return cov ? (μt, Tuple(v[:, l] for l in 1:n)) : μt
end

# predict_y predictions.jl:178-198: regression mean / Bool / most likely class label / expected count
function predict_y(hm::HipModel{T}, X_test::AbstractMatrix, state=nothing; obsdim::Int=1) where {T}
    ensure_handle!(hm, max(hm.maxbatch, 1))
    Xd = ROCArray{T}(obsdim == 1 ? permutedims(X_test) : X_test)
    D, nt = size(Xd)
    l = AGP.likelihood(hm.model)
    if is_mo(hm)
        out = ROCMatrix{T}(undef, nt, length(l))
        check(hm.ctx, ccall((:agp_svgp_predict_y, libagp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}), hm.h, pointer(Xd), D, nt, pointer(out)))
        o = Array(out)
        return [is_bernoulli(lt) ? o[:, t] .> 0.5 : o[:, t] for (t, lt) in enumerate(l)]
    elseif is_bernoulli(l) || is_multiclass(l)
        out = ROCVector{Int32}(undef, nt)
        check(hm.ctx, ccall((:agp_svgp_predict_y, libagp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}), hm.h, pointer(Xd), D, nt, pointer(out)))
        o = Array(out)
        return is_bernoulli(l) ? o .== 1 : [l.class_mapping[i + 1] for i in o]     # multiclass.jl:96-99 / predictions.jl:200-202
    else
        out = ROCVector{T}(undef, nt)
        check(hm.ctx, ccall((:agp_svgp_predict_y, libagp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}), hm.h, pointer(Xd), D, nt, pointer(out)))
        return Array(out)
    end
end

# proba_y predictions.jl:225-247 + compute_proba: (mean, var) for regression, p for Bernoulli, per-class probabilities
function proba_y(hm::HipModel{T}, X_test::AbstractMatrix, state=nothing; obsdim::Int=1) where {T}
    ensure_handle!(hm, max(hm.maxbatch, 1))
    Xd = ROCArray{T}(obsdim == 1 ? permutedims(X_test) : X_test)
    D, nt = size(Xd)
    l = AGP.likelihood(hm.model)
    nout = is_mo(hm) ? length(l) : 1
    o0 = is_multiclass(l) ? ROCMatrix{T}(undef, nlat(hm), nt) : ROCMatrix{T}(undef, nt, nout)
    o1 = is_multiclass(l) ? nothing : ROCMatrix{T}(undef, nt, nout)
    check(hm.ctx, ccall((:agp_svgp_proba_y, libagp), Int32,
                        (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{Cvoid}, Ptr{Cvoid}),
                        hm.h, pointer(Xd), D, nt, AGP.pred_nodes, AGP.pred_weights, length(AGP.pred_nodes), pointer(o0),
                        o1 === nothing ? C_NULL : pointer(o1)))
    ccall((:agp_ctx_sync, libagp), Int32, (Ptr{Cvoid},), hm.ctx)
    if is_multiclass(l)
        p = Array(o0)                                        # K x n_t (ABI: T[n_t][K] row-major)
        return NamedTuple{Tuple(Symbol.(l.class_mapping))}(Tuple(p[k, :] for k in 1:size(p, 1)))   # multiclass.jl:101-117
    end
    a, b = Array(o0), Array(o1)
    is_mo(hm) && return [(a[:, t], b[:, t]) for t in 1:nout]
    return is_bernoulli(l) ? a[:, 1] : (a[:, 1], b[:, 1])
end

# ---- multi-GPU: one Julia process per GPU ---------------------------------------------------------------------------------------
# rank 0:  id = AGPHip.comm_unique_id()  -> broadcast the 128 bytes (MPI.Bcast!, Distributed, a file ...)
# all:     AGPHip.comm_init!(hm, rank, world, id; shard=:batch | :latent)
# then train! / objective / update_hyperparameters! route through the *_multi entry points; RCCL runs on the model's stream.
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    st = ccall((:agp_comm_unique_id, libagp), Int32, (Ptr{UInt8},), id)
    st == 0 || throw(AGPError(st, "agp_comm_unique_id (librccl not found? set AGP_RCCL_PATH)"))
    return id
end
function comm_init!(hm::HipModel, rank::Integer, world::Integer, id::Vector{UInt8}; shard::Symbol=:batch)
    length(id) == 128 || throw(ArgumentError("the RCCL id has 128 bytes"))
    c = Ref{Ptr{Cvoid}}()
    check(ensure_ctx!(hm), ccall((:agp_comm_init, libagp), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{UInt8}, Ptr{Ptr{Cvoid}}),
                                 hm.ctx, rank, world, id, c))
    hm.comm = c[]
    hm.shard = shard === :batch ? AGP_SHARD_BATCH : AGP_SHARD_LATENT
    hm.rank, hm.world = Int32(rank), Int32(world)            # ensure_handle! re-applies them to every handle it creates later
    if shard === :batch && hm.h != C_NULL
        check(hm.ctx, ccall((:agp_svgp_set_batch_shard, libagp), Int32, (Ptr{Cvoid}, Int32, Int32), hm.h, rank, world))
    end
    return hm
end
function comm_info(hm::HipModel)
    r, w, k = Ref{Int32}(), Ref{Int32}(), Ref{Int32}()
    ccall((:agp_comm_info, libagp), Int32, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}, Ref{Int32}), hm.comm, r, w, k)
    return Int(r[]), Int(w[]), k[] == 1
end
comm_rank(hm::HipModel) = comm_info(hm)[1]
comm_world(hm::HipModel) = comm_info(hm)[2]
# latent-parallel: build every rank's twin with HipModel(model; latent_range = latent_slice(length(model.f), world, rank))
function latent_slice(n::Int, world::Int, rank::Int)
    base, rem = divrem(n, world)
    lo = rank * base + min(rank, rem)
    return (lo + 1):(lo + base + (rank < rem ? 1 : 0))
end

# ---- inducing-point selection on the device (InducingPoints.jl's interface, agp_kmeans behind it) --------------------------------
# `inducingpoints(KmeansAlg(m), X)` is how every example / test of the reference picks Z (test/testingtools.jl:66,
# docs/examples/gpclassification.jl:47).  The random part (AFK-MC² seeding: first centre, proposals, Metropolis acceptances) needs
# the caller's RNG and stays on the host, on a few thousand gathered candidates; the O(N m D) distance passes and the Lloyd
# iterations run on the GPU (csrc/agp_kmeans.h).  Returns the same `Vector{Vector{T}}` the reference's constructors take.
function inducingpoints(alg::AGP.KmeansAlg, X::AbstractMatrix{T}; obsdim::Int=1, rng=Random.default_rng(),
                        nMarkov::Int=10, tol::Real=1e-3, maxiter::Int=100) where {T<:Union{Float32,Float64}}
    Xd = ROCArray{T}(obsdim == 1 ? permutedims(X) : X)        # D x N == point-major
    D, N = size(Xd); m = alg.m
    m <= N || error("Input data not big enough given the desired number of inducing points")
    ctx = Ref{Ptr{Cvoid}}()
    st = ccall((:agp_ctx_create, libagp), Int32, (Int32, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), AMDGPU.device_id() - 1, AMDGPU.stream().stream, ctx)
    st == 0 || throw(AGPError(st, "agp_ctx_create"))
    dt = T == Float64 ? Int32(0) : Int32(1)
    try
        first = rand(rng, 1:N)
        c1 = Xd[:, first:first]
        d1 = ROCVector{T}(undef, N)
        check(ctx[], ccall((:agp_nearest_center, libagp), Int32,
                           (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ptr{Int32}, Ptr{Cvoid}),
                           ctx[], dt, pointer(Xd), N, D, D, pointer(c1), D, 1, C_NULL, pointer(d1)))
        q = Float64.(Array(d1)); q = q ./ sum(q) ./ 2 .+ 1 / (2N); q ./= sum(q)
        seeds = Matrix{Float64}(undef, D, m); seeds[:, 1] = Array(c1)
        Xh = nothing
        for i in 2:m                                           # one short Metropolis chain per further centre
            prop = sample(rng, 1:N, Weights(q), nMarkov)
            cand = Float64.(Array(Xd[:, prop]))
            dmin(v) = minimum(sum(abs2, seeds[:, 1:(i - 1)] .- v; dims=1))
            x = cand[:, 1]; dx = dmin(x)
            for j in 2:nMarkov
                yv = cand[:, j]; dy = dmin(yv)
                if dy > rand(rng) * dx
                    x, dx = yv, dy
                end
            end
            seeds[:, i] = x
        end
        Cd = ROCArray{T}(seeds)
        it, conv, obj = Ref{Int32}(), Ref{Int32}(), Ref{Float64}()
        check(ctx[], ccall((:agp_kmeans, libagp), Int32,
                           (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Int32, Float64, Ptr{Int32},
                            Ptr{Int32}, Ref{Int32}, Ref{Float64}, Ref{Int32}),
                           ctx[], dt, pointer(Xd), N, D, D, pointer(Cd), D, m, maxiter, tol, C_NULL, C_NULL, it, obj, conv))
        Z = Array(Cd)
        return [Z[:, j] for j in 1:m]
    finally
        ccall((:agp_ctx_destroy, libagp), Int32, (Ptr{Cvoid},), ctx[])
    end
end

# ---- OnlineSVGP: streaming batches (src/models/OnlineSVGP.jl, src/training/onlinetraining.jl:17-218) ---------------------------
# A streaming model is a CHAIN of device handles: every arriving batch may grow Z (the reference's own InducingPoints.updateZ /
# remove_point run on the host, exactly where onlinetraining.jl:153-172 calls them), so a fresh handle is created for the new
# inducing points and the previous posterior is installed as its prior:
#   save_old_gp! (:170-180)          -> agp_svgp_online_snapshot  (invDₐ = Σₐ⁻¹ − Kₐ⁻¹, η₁ₐ, 𝓛ₐ read off the old handle)
#   compute_kernel_matrices (:199-237) -> agp_svgp_set_online_prior (K_ab, κₐ, K̃ₐ formed at the next K refresh)
#   first iteration (:78-104)        -> agp_svgp_online_first_step (local update under the OLD inducing points / posterior)
#   later iterations                 -> agp_svgp_cavi_step (full batch, closed-form natural parameters, analyticVI.jl:183-203)
# Only AnalyticVI() is accepted (the reference's stochastic branch uses an undefined name, onlinetraining.jl:52).  All latents of a
# handle share m: latents whose OIPS runs end with fewer points are filled with neutral far-away points (every kernel value that
# involves one is exactly 0), as the Python mirror does (augmentedgaussianprocesses.jl_amd/online.py, _pad_inducing).
mutable struct HipOnlineModel{T}
    model::AGP.OnlineSVGP{T}
    cur::Union{Nothing,HipModel}     # device twin of the current inducing points (an SVGP view of model.f)
    m_real::Vector{Int}
end
HipOnlineModel(m::AGP.OnlineSVGP{T}) where {T} = HipOnlineModel{T}(m, nothing, Int[])

const FAR = 1.0e6
function pad_inducing(Zs::Vector{<:AbstractVector})
    mmax = maximum(length, Zs)
    D = length(first(first(Zs)))
    return [vcat(collect.(Z), [fill(FAR * j, D) for j in 1:(mmax - length(Z))]) for Z in Zs], length.(Zs)
end

function train!(ho::HipOnlineModel{T}, X::AbstractMatrix, y::AbstractArray, state=nothing; iterations::Int=20,
                callback=nothing, obsdim::Int=1) where {T}
    iterations > 0 || error("Number of iterations should be positive")
    m = ho.model
    AGP.is_stochastic(m) && error("OnlineSVGP streams full batches on this path (onlinetraining.jl:48-53 cannot run)")
    Xv = KernelFunctions.vec_of_vecs(X; obsdim)
    first_batch = AGP.n_iter(m) == 0
    old = ho.cur
    snaps = nothing
    if first_batch
        AGP.init_online_model(m, Xv)                                          # onlinetraining.jl:182-197 (OIPS on the host)
    else
        pull_hypers!(old)                                                     # kernels / Z as the last hyper steps left them
        snaps = map(1:nlat(old)) do i                                         # save_old_gp!, :170-180
            mo = old.maxbatch; mm = AGP.dim(old.model.f[i])
            iD = ROCMatrix{T}(undef, mm, mm); e1 = ROCVector{T}(undef, mm); pl = Ref{Float64}()
            check(old.ctx, ccall((:agp_svgp_online_snapshot, libagp), Int32,
                                 (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ref{Float64}), old.h, i - 1, pointer(iD), mm, pointer(e1), pl))
            (iD, e1, pl[], reduce(hcat, AGP.Zview(old.model.f[i])))
        end
        for gp in m.f                                                         # remove_point (:172) then updateZ (:153-160)
            gp.Zₐ = deepcopy(gp.Z)
            gp.Z = AGP.InducingPoints.remove_point(Random.GLOBAL_RNG, gp.Z, gp.Zalg,
                                                   kernelmatrix(AGP.kernel(gp), gp.Z) + T(AGP.jitt) * I)
            gp.Z = AGP.InducingPoints.updateZ(gp.Z, gp.Zalg, Xv; kernel=AGP.kernel(gp))
            gp.post.dim = length(AGP.Zview(gp))
        end
    end
    Zs, m_real = pad_inducing([AGP.Zview(gp) for gp in m.f])
    # an SVGP view of the online latents (same kernels / optimisers, padded Z): what the device handle is created from
    view = SVGP(AGP.kernel.(m.f) |> first, AGP.likelihood(m), AnalyticVI(), Zs[1]; optimiser=AGP.opt(first(m.f)),
                Zoptimiser=AGP.Zopt(first(m.f)), atfrequency=m.atfrequency)
    for (gp, k, Z) in zip(view.f, AGP.kernel.(m.f), Zs)
        gp.prior.kernel = k; gp.Z = Z
    end
    new = HipModel(view)
    data = upload!(new, X, y; obsdim)
    B = new.N
    ensure_handle!(new, max(B, old === nothing ? 0 : old.maxbatch))
    mp = length(Zs[1])
    for i in 1:length(m.f)
        if first_batch                                                        # init_opt_state(::OnlineVarLatent), states.jl:85-97
            E = Matrix{T}(I, mp, mp); E[(m_real[i] + 1):end, (m_real[i] + 1):end] .= 0
            Ed = ROCArray(E); z = ROCArray(zeros(T, mp))
            check(new.ctx, ccall((:agp_svgp_set_online_prior, libagp), Int32,
                                 (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Float64),
                                 new.h, i - 1, C_NULL, 0, mp, pointer(Ed), mp, pointer(z), 0.0))
        else
            iD, e1, pl, Za = snaps[i]
            Zad = ROCArray{T}(Za)
            check(new.ctx, ccall((:agp_svgp_set_online_prior, libagp), Int32,
                                 (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Float64),
                                 new.h, i - 1, pointer(Zad), size(Za, 1), size(Za, 2), pointer(iD), size(iD, 1), pointer(e1), pl))
        end
    end
    start = 1
    if !first_batch      # first iteration: local update under the old inducing points, natural gradient under the new (:78-104)
        check(new.ctx, ccall((:agp_svgp_online_first_step, libagp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64),
                             new.h, old.h, pointer(new.X), size(new.X, 1), pointer(new.y), B))
        start = 2
    end
    hyper_on = any(gp -> AGP.opt(gp) !== nothing || AGP.Zopt(gp) !== nothing, m.f)
    for it in 1:iterations
        it >= start && update_parameters!(new, nothing, 1.0)
        callback === nothing || callback(ho, new, AGP.n_iter(m))
        if hyper_on && AGP.n_iter(m) % m.atfrequency == 0 && AGP.n_iter(m) >= 3                     # :112-114
            update_hyperparameters!(new)
        end
        AGP.inference(m).n_iter += 1
    end
    check(new.ctx, ccall((:agp_svgp_check_status, libagp), Int32, (Ptr{Cvoid},), new.h))
    check(new.ctx, ccall((:agp_svgp_refresh_K, libagp), Int32, (Ptr{Cvoid},), new.h))
    pull_posterior!(new); pull_hypers!(new)
    for (gp, vgp, n) in zip(m.f, view.f, m_real)                              # back into the reference object, padding stripped
        gp.post.μ = vgp.post.μ[1:n]; gp.post.Σ = Symmetric(Matrix(vgp.post.Σ)[1:n, 1:n])
        gp.post.η₁ = vgp.post.η₁[1:n]; gp.post.η₂ = Symmetric(Matrix(vgp.post.η₂)[1:n, 1:n])
        gp.Z = vgp.Z[1:n]
    end
    old === nothing || destroy!(old)
    ho.cur, ho.m_real = new, m_real
    AGP.set_trained!(m, true)
    return ho, new
end
predict_f(ho::HipOnlineModel, Xt::AbstractMatrix, state=nothing; kw...) = predict_f(ho.cur, Xt; kw...)
predict_y(ho::HipOnlineModel, Xt::AbstractMatrix, state=nothing; kw...) = predict_y(ho.cur, Xt; kw...)
proba_y(ho::HipOnlineModel, Xt::AbstractMatrix, state=nothing; kw...) = proba_y(ho.cur, Xt; kw...)
objective(ho::HipOnlineModel, state=nothing, y=nothing) = objective(ho.cur)   # incl. −extraKL (KLdivergences.jl:30-54)

# ---- drop-in methods on the reference's own types ----------------------------------------------------------------------------------
# With the shim loaded, `train!(model, X, y, n)` on an SVGP / MOSVGP / OnlineSVGP with AnalyticVI / AnalyticSVI runs on the device:
# these methods are MORE SPECIFIC than the reference's `train!(model::AbstractGPModel, X::AbstractArray, y::AbstractArray, ...)`
# (src/training/training.jl:13-22) and `update_parameters!` entry (:140-144), so dispatch selects them -- no wrapper call at the
# user's site; `backend=:cpu` (or `AGPHip.default_backend!(:cpu)`) falls through to the reference's own method with `invoke`.
# The device twin of a model lives in a WeakKeyDict keyed by the model object; predictions, ELBO and objective look it up and fall
# back to the reference when the model was never trained on the device.  (As a package extension this section is
# `ext/AGPHipExt.jl`, triggered by `using AMDGPU`.)
const HipSVGP{T} = Union{SVGP{T,<:Any,<:AnalyticVI},AGP.MOSVGP{T,<:Any,<:AnalyticVI}}
const BACKEND = Ref(:hip)
default_backend!(b::Symbol) = (b in (:hip, :cpu) || throw(ArgumentError("backend is :hip or :cpu")); BACKEND[] = b)
const TWINS = WeakKeyDict{Any,Any}()
twin(model::HipSVGP; kw...) = get!(() -> HipModel(model; kw...), TWINS, model)
twin(model::AGP.OnlineSVGP) = get!(() -> HipOnlineModel(model), TWINS, model)
has_twin(model) = haskey(TWINS, model)

function train!(model::HipSVGP, X::AbstractArray, y::AbstractArray, iterations::Int=100; backend::Symbol=BACKEND[],
                reference_compat_stale_K::Bool=false, kwargs...)
    backend === :cpu &&
        return invoke(train!, Tuple{AGP.AbstractGPModel,AbstractArray,AbstractArray,Int}, model, X, y, iterations; kwargs...)
    hm = twin(model; reference_compat_stale_K)
    _, state = train!(hm, X, y, iterations; kwargs...)      # leaves the trained posterior / kernels / Z in `model` as well
    return model, state
end
function train!(model::AGP.OnlineSVGP{T,<:Any,<:AnalyticVI}, X::AbstractMatrix, y::AbstractArray, state=nothing;
                backend::Symbol=BACKEND[], kwargs...) where {T}
    backend === :cpu &&
        return invoke(train!, Tuple{AGP.OnlineSVGP,AbstractMatrix,AbstractArray,Any}, model, X, y, state; kwargs...)
    _, st = train!(twin(model), X, y, state; kwargs...)
    return model, st
end
for f in (:predict_f, :predict_y, :proba_y)
    @eval function $f(model::Union{HipSVGP,AGP.OnlineSVGP}, X_test::AbstractMatrix, state=nothing; backend::Symbol=BACKEND[], kw...)
        (backend === :cpu || !has_twin(model)) &&
            return invoke($f, Tuple{AGP.AbstractGPModel,AbstractMatrix,Any}, model, X_test, state; kw...)
        return $f(TWINS[model], X_test; kw...)
    end
end
function ELBO(model::HipSVGP, X::AbstractMatrix, y::AbstractArray; backend::Symbol=BACKEND[], kw...)
    (backend === :cpu || !has_twin(model)) && return invoke(ELBO, Tuple{AGP.AbstractGPModel,AbstractMatrix,AbstractArray}, model, X, y; kw...)
    return ELBO(TWINS[model], X, y; kw...)
end
objective(model::HipSVGP, state::HipModel, y=nothing) = objective(state)    # the state train! returned IS the device twin

end # module
