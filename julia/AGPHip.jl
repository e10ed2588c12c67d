# AGPHip.jl -- thin Julia shim over libagp_hip.so (include/agp_hip.h).
#
# NOT EXECUTED IN THE BUILD ENVIRONMENT (no julia binary there); it documents, in the reference's own language, the
# exact binding a maintainer of AugmentedGaussianProcesses.jl would add so that SVGP / AnalyticVI / AnalyticSVI models
# run their hot path on an MI355X.  AMDGPU.jl is used ONLY for device / stream / buffer handles (ROCArray, stream
# pointer); there is no KernelAbstractions and no CUDA.jl compatibility layer: every numeric operation is a `ccall`
# into hand-written HIP.
#
# Seams replaced (SURVEY.md section 8b):
#   update_parameters!(model::SVGP, state, x, y)          src/training/training.jl:140-144  -> agp_svgp_cavi_step
#   compute_K / compute_κ                                  src/gpblocks/latentgp.jl:205-215  -> inside cavi_step / agp_svgp_refresh_K
#   ELBO(model, state, y) / ELBO(model, X, y)              src/inference/analyticVI.jl:255-274, src/functions/ELBO.jl:32-47 -> agp_svgp_elbo
#   _predict_f / predict_y / proba_y                       src/training/predictions.jl:25-50,178-247 -> agp_svgp_predict_f / _predict_y / _proba_y
module AGPHip

using AMDGPU
using AugmentedGaussianProcesses
using KernelFunctions
const AGP = AugmentedGaussianProcesses

const libagp = joinpath(@__DIR__, "..", "augmentedgaussianprocesses.jl_amd", "libagp_hip.so")

# ---- POD mirrors of include/agp_hip.h ---------------------------------------------------------------------------
struct KernelDesc
    kind::Int32
    ard::Int32
    variance::Float64
    scale::Float64
    ard_scales_host::Ptr{Float64}
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
    reserved::Int32
end

struct AGPError <: Exception
    status::Int32
    msg::String
end
function check(ctx, st::Int32)
    st == 0 && return nothing
    msg = unsafe_string(ccall((:agp_last_error, libagp), Cstring, (Ptr{Cvoid},), ctx))
    st == 2 && throw(PosDefException(0))                       # cholesky failure, latentgp.jl:206
    st == 3 && error("K̃ has negative values")                  # latentgp.jl:213
    throw(AGPError(st, msg))
end

# ---- KernelFunctions.jl objects -> agp_kernel_desc ----------------------------------------------------------------
kind(::SqExponentialKernel) = Int32(0)
kind(::Matern52Kernel) = Int32(1)
kind(::Matern32Kernel) = Int32(2)
kind(::ExponentialKernel) = Int32(3)
function kernel_desc(k::Kernel, D::Int)
    σ² = 1.0
    if k isa ScaledKernel
        σ² = only(k.σ²); k = k.kernel
    end
    if k isa TransformedKernel
        t = k.transform; b = k.kernel
        if t isa ScaleTransform
            return KernelDesc(kind(b), 0, σ², only(t.s), C_NULL), nothing
        elseif t isa ARDTransform
            v = Vector{Float64}(t.v)
            return KernelDesc(kind(b), 1, σ², 1.0, pointer(v)), v      # keep v alive (GC.@preserve at the call)
        end
    end
    return KernelDesc(kind(k), 0, σ², 1.0, C_NULL), nothing
end

lik_desc(l::GaussianLikelihood) = LikDesc(0, 1, AGP.noise(l), 0.0)
lik_desc(::AGP.BernoulliLikelihood{<:AGP.LogisticLink}) = LikDesc(1, 1, 0.0, 0.0)
lik_desc(l::StudentTLikelihood) = LikDesc(2, 1, l.ν, l.σ)
lik_desc(l::AGP.MultiClassLikelihood{<:AGP.LogisticSoftMaxLink}) = LikDesc(3, AGP.n_class(l), 0.0, 0.0)
lik_desc(l::LaplaceLikelihood) = LikDesc(5, 1, l.β, 0.0)
lik_desc(::AGP.BernoulliLikelihood{<:AGP.SVMLink}) = LikDesc(6, 1, 0.0, 0.0)
lik_desc(l::AGP.PoissonLikelihood{<:AGP.ScaledLogistic}) = LikDesc(7, 1, only(l.invlink.λ), 0.0)   # λ is state: see pull_lik_state!
lik_desc(l::NegBinomialLikelihood) = LikDesc(8, 1, Float64(l.r), 0.0)
lik_desc(l::AGP.HeteroscedasticGaussianLikelihood{<:AGP.InvScaledLogistic}) = LikDesc(9, 1, only(l.invlink.λ), 0.0)

# λ of Poisson / Heteroscedastic is re-estimated on the device by every local update (poisson.jl:78, heteroscedastic.jl:95);
# copy it back into the reference object after training / before predicting with reference-side code.
function pull_lik_state!(s, l::Union{AGP.PoissonLikelihood,AGP.HeteroscedasticGaussianLikelihood})
    v = Ref{Float64}()
    check(s.ctx, ccall((:agp_svgp_get_lik_param, libagp), Int32, (Ptr{Cvoid}, Ref{Float64}), s.h, v))
    l.invlink.λ .= v[]
    return l
end
pull_lik_state!(s, l) = l

# ---- device handle living next to the reference model -----------------------------------------------------------------
mutable struct HipState{T}
    ctx::Ptr{Cvoid}
    h::Ptr{Cvoid}
    X::ROCMatrix{T}          # D x N (ColVecs memory order == the ABI's point-major layout)
    y::ROCVector              # T (±1 / real) or Int32 class index
    maxbatch::Int
end

function make_handle(model::SVGP{T}, X::AbstractMatrix, y, maxbatch::Int; obsdim=1) where {T}
    ctx = Ref{Ptr{Cvoid}}()
    stream = AMDGPU.stream().stream                         # share AMDGPU.jl's HIP stream with the library
    st = ccall((:agp_ctx_create, libagp), Int32, (Int32, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), AMDGPU.device_id() - 1, stream, ctx)
    st == 0 || throw(AGPError(st, "agp_ctx_create"))
    Xd = ROCArray{T}(obsdim == 1 ? permutedims(X) : X)      # one-time RowVecs -> point-major permutation
    inf = AGP.inference(model)
    D, N = size(Xd)
    m = AGP.dim(model.f[1])
    desc = SvgpDesc(T == Float64 ? 0 : 1, AGP.n_latent(model), 0, AGP.is_stochastic(inf) ? 1 : 0, m, D, maxbatch,
                    lik_desc(AGP.likelihood(model)), 0.0,
                    AGP.is_stochastic(inf) ? inf.vi_opt.optimiser.κ : 0.51, AGP.is_stochastic(inf) ? inf.vi_opt.optimiser.τ : 1.0,
                    0, 0)
    h = Ref{Ptr{Cvoid}}()
    check(ctx[], ccall((:agp_svgp_create, libagp), Int32, (Ptr{Cvoid}, Ref{SvgpDesc}, Ptr{Ptr{Cvoid}}), ctx[], desc, h))
    for (i, gp) in enumerate(model.f)
        kd, keep = kernel_desc(AGP.kernel(gp), D)
        GC.@preserve keep check(ctx[], ccall((:agp_svgp_set_kernel, libagp), Int32, (Ptr{Cvoid}, Int32, Ref{KernelDesc}), h[], i - 1, kd))
        Zd = ROCArray{T}(reduce(hcat, AGP.Zview(gp)))       # D x m, point-major
        check(ctx[], ccall((:agp_svgp_set_Z, libagp), Int32, (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Int64), h[], i - 1, pointer(Zd), D))
    end
    if AGP.likelihood(model) isa AGP.PoissonLikelihood       # the λ update integrates logistic by Gauss-Hermite (utils.jl:16-19)
        check(ctx[], ccall((:agp_svgp_set_quadrature, libagp), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32),
                           h[], AGP.pred_nodes, AGP.pred_weights, length(AGP.pred_nodes)))
    end
    yd = AGP.likelihood(model) isa AGP.MultiClassLikelihood ? ROCArray(Int32.(map(r -> findfirst(r) - 1, eachrow(y)))) : ROCArray{T}(y)
    return HipState{T}(ctx[], h[], Xd, yd, maxbatch)
end

# ---- update_parameters!(model::SVGP, state, x, y) replacement (training.jl:140-144) ------------------------------------
# `idx` is the minibatch drawn by train! (StatsBase.sample, training.jl:51-53); it is uploaded instead of a view of X.
function update_parameters_hip!(s::HipState{T}, idx::Union{Nothing,Vector{Int}}, ρ::Real) where {T}
    D = size(s.X, 1)
    if idx === nothing
        B = size(s.X, 2); idp = C_NULL
        st = ccall((:agp_svgp_cavi_step, libagp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Int64}, Int64, Float64),
                   s.h, pointer(s.X), D, pointer(s.y), idp, B, ρ)
    else
        idd = ROCArray(Int64.(idx .- 1))                     # 0-based on the device
        st = ccall((:agp_svgp_cavi_step, libagp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Int64}, Int64, Float64),
                   s.h, pointer(s.X), D, pointer(s.y), pointer(idd), length(idx), ρ)
    end
    check(s.ctx, st)
end

# pull (μ, Σ, η₁, η₂) back into the reference's VarPosterior (posterior.jl:21-27), e.g. at the end of train!
function sync_posterior!(model::SVGP{T}, s::HipState{T}) where {T}
    for (i, gp) in enumerate(model.f)
        m = AGP.dim(gp)
        μ = ROCVector{T}(undef, m); η₁ = ROCVector{T}(undef, m)
        Σ = ROCMatrix{T}(undef, m, m); η₂ = ROCMatrix{T}(undef, m, m)
        check(s.ctx, ccall((:agp_svgp_get_state, libagp), Int32, (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                           s.h, i - 1, pointer(μ), pointer(Σ), pointer(η₁), pointer(η₂)))
        ccall((:agp_ctx_sync, libagp), Int32, (Ptr{Cvoid},), s.ctx)
        gp.post.μ .= Array(μ); gp.post.η₁ .= Array(η₁)
        gp.post.Σ.data .= Array(Σ); gp.post.η₂.data .= Array(η₂)   # symmetric: row-major == column-major
    end
    return model
end

# predict_f(model, X_test; cov) replacement (predictions.jl:25-50): streams over the test points on the device
function predict_f_hip(s::HipState{T}, Xt::AbstractMatrix; cov::Bool=false, obsdim=1, n_latent=1) where {T}
    Xd = ROCArray{T}(obsdim == 1 ? permutedims(Xt) : Xt)
    D, nt = size(Xd)
    μ = ROCMatrix{T}(undef, nt, n_latent)                  # column l == latent l  (ABI: T[n_latent][n_t])
    v = cov ? ROCMatrix{T}(undef, nt, n_latent) : nothing
    check(s.ctx, ccall((:agp_svgp_predict_f, libagp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                       s.h, pointer(Xd), D, nt, pointer(μ), cov ? pointer(v) : C_NULL))
    ccall((:agp_ctx_sync, libagp), Int32, (Ptr{Cvoid},), s.ctx)
    return cov ? (Array(μ), Array(v)) : Array(μ)
end

end # module
