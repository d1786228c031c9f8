# IIFNbpExt.jl -- the Julia side of the libnbp boundary (include/nbp.h, include/nbp_host.h).
#
# A package extension a maintainer of IncrementalInference.jl adds (Project.toml: [extensions] IIFNbpExt = "..."):
# methods that are MORE SPECIFIC than the generic ones for the factor types libnbp implements, forwarding to
# `libnbp.so` with `ccall`.  The CliqueStateMachine, the Bayes tree, DistributedFactorGraphs and message passing stay
# in Julia; per clique the CSM makes the two calls it makes today,
#     upGibbsCliqueDensity(dfg, cliq, solveKey, inmsgs, N, dbg, iters, logger)   src/services/SolveTree.jl:164-239
#     solveCliqDownFrontalProducts!(subfg, cliq, opts, logger)                    CliqStateMachineUtils.jl:479-571
# and each becomes ONE ccall (nbp_clique_upsolve / nbp_clique_downsolve).
#
# Julia is not part of the build image of libnbp, so this file has not been executed there.  What IS machine-checked
# is the part that silently corrupts memory when it drifts: every `struct` below is parsed by
# tests/test_julia_shim_layout.py and compared, field by field (name, offset, size), with what gcc computes from the C
# headers.  Keep one field per `name::Type` and C field names.
module IIFNbpExt

using IncrementalInference
using DistributedFactorGraphs
using ApproxManifoldProducts
using Manifolds
using StaticArrays
using RecursiveArrayTools: ArrayPartition
using Distributions
using LinearAlgebra
using Logging: ConsoleLogger

# Dispatch contract of the overrides below.  None of them repeats a reference signature (a method with the signature of the
# reference's would OVERWRITE it -- an error while an extension precompiles on Julia >= 1.10, and the `invoke` fall-backs
# would then reach the override itself and recurse).  Every override is strictly MORE SPECIFIC in its graph argument: the
# clique sub graph a state machine works on is an in-memory graph (`CliqStateMachineContainer.cliqSubFg::InMemG`,
# entities/JunctionTreeTypes.jl:39; `LocalDFG` = `GraphsDFG`), whereas the reference's methods take `AbstractDFG`
# (SolveTree.jl:164, CliqStateMachineUtils.jl:479, TreeMessageUtils.jl:279).  `invoke(f, Tuple{AbstractDFG, ...}, ...)`
# therefore reaches the reference's method; a graph of another type never enters the shim.
import IncrementalInference: upGibbsCliqueDensity, solveCliqDownFrontalProducts!, approxConvBelief, addLikelihoodsDifferentialCHILD!,
                             getSolverParams, getVariableType, getFactorType, getCliqueData, _getCCW,
                             TreeBelief, TreeClique, MsgPrior, setValKDE!

const libnbp = get(ENV, "LIBNBP", "libnbp.so")

# ---- constants and enums of include/nbp.h -------------------------------------------------------------------
const NBP_MAXV = 6
const NBP_MAXF = 128
const NBP_MAXC = 4
const NBP_COMP_STRIDE = 13

const NBP_EUCLID1, NBP_EUCLID2, NBP_EUCLID3, NBP_CIRCULAR, NBP_SE2 = Int32(1), Int32(2), Int32(3), Int32(4), Int32(5)
const NBP_F_PRIOR, NBP_F_MSGPRIOR, NBP_F_LINREL, NBP_F_CIRCULAR, NBP_F_SE2, NBP_F_EUCLIDDIST, NBP_F_PASSTHROUGH =
  Int32(1), Int32(2), Int32(3), Int32(4), Int32(5), Int32(6), Int32(7)
const NBP_SOLVER_STORED_MEASUREMENTS, NBP_SOLVER_MSG_LIKELIHOODS = Int32(1), Int32(2)

# ---- structs: byte-for-byte mirrors (checked by tests/test_julia_shim_layout.py) ----------------------------------
# nbp_proposal_desc (include/nbp.h): one approxConvBelief
struct NbpProposalDesc
  factor_kind::Int32
  manifold::Int32
  nvars::Int32
  sfidx::Int32
  var_slot::NTuple{NBP_MAXV, Int32}
  out_slot::Int32
  ncomp::Int32
  has_multihypo::Int32
  inflate_cycles::Int32
  mhidx_in::Int32
  mhidx_out::Int32
  skip_bandwidth::Int32
  partial_mask::Int32
  meas_kde::Int32
  keep_count::Int32
  multihypo::NTuple{NBP_MAXV, Float64}
  nullhypo::Float64
  inflation::Float64
  spread_nh::Float64
  comp::NTuple{NBP_MAXC * NBP_COMP_STRIDE, Float64}
  seed::UInt64
  meas_seed::UInt64
end

# nbp_product_desc (include/nbp.h): one AMP.manifoldProduct + rebandwidth
struct NbpProductDesc
  manifold::Int32
  nfactors::Int32
  niter::Int32
  out_slot::Int32
  in_slot::NTuple{NBP_MAXF, Int32}
  labels_out::Int32
  old_slot::Int32
  in_partial::NTuple{NBP_MAXF, UInt8}
  seed::UInt64
end

# nbp_copy_desc
struct NbpCopyDesc
  src_slot::Int32
  dst_slot::Int32
end

# nbp_diag
struct NbpDiag
  solves::Int64
  nonconverged::Int64
  nan_results::Int64
  residual_evals::Int64
  lcv_evals::Int64
  lcv_evals_f32::Int64
end

# nbp_solver_params (include/nbp_host.h): the SolverParams fields the path reads (entities/SolverParams.jl:12-75)
struct NbpSolverParams
  N::Int32
  gibbs_iters::Int32
  inflate_cycles::Int32
  product_niter::Int32
  upsolve::Int32
  downsolve::Int32
  limitfixeddown::Int32
  flags::Int32
  spread_nh::Float64
  inflation::Float64
  null_surplus_add::Float64
end

# nbp_factor_spec (include/nbp_host.h): addFactor!(dfg, Xi, usrfnc; multihypo, nullhypo, inflation)
struct NbpFactorSpec
  factor_kind::Int32
  nvars::Int32
  vars::NTuple{NBP_MAXV, Int32}
  ncomp::Int32
  has_multihypo::Int32
  partial_mask::Int32
  multihypo::NTuple{NBP_MAXV, Float64}
  nullhypo::Float64
  inflation::Float64
  comp::NTuple{NBP_MAXC * NBP_COMP_STRIDE, Float64}
end

# nbp_tree_belief (include/nbp_host.h): TreeBelief (val, bw, infoPerCoord) in caller-owned buffers
struct NbpTreeBelief
  pts::Ptr{Float64}
  bw::Ptr{Float64}
  ipc::Ptr{Float64}
  n_pts::Int32
  handle::Int32       # 0: host buffers; h > 0: resident slot h of the context (nbp_ctx_reserve_resident)
end

# nbp_clique_desc (include/nbp_host.h)
struct NbpCliqueDesc
  clique_id::Int32
  nvars::Int32
  nfrontals::Int32
  nseparators::Int32
  manifold::Ptr{Int32}
  ismargin::Ptr{Int32}
  nfactors::Int32
  factors::Ptr{NbpFactorSpec}
  n_direct_frtl_msg::Int32
  n_msgskip::Int32
  n_itervar::Int32
  n_direct_prior_msg::Int32
  direct_frtl_msg::Ptr{Int32}
  msgskip::Ptr{Int32}
  itervar::Ptr{Int32}
  direct_prior_msg::Ptr{Int32}
  nmsgs::Int32
  msg_var::Ptr{Int32}
  msg_belief::Ptr{NbpTreeBelief}
  factor_density::Ptr{NbpTreeBelief}
  factor_meas_kde::Ptr{NbpTreeBelief}
  n_diff::Int32
  reserved_::Int32
  diff_a::Ptr{Int32}
  diff_b::Ptr{Int32}
  diff_kind::Ptr{Int32}
end

# nbp_clique_request (include/nbp_host.h): one entry of nbp_clique_solve_batch
struct NbpCliqueRequest
  params::Ptr{NbpSolverParams}
  clique::Ptr{NbpCliqueDesc}
  seed::UInt64
  beliefs::Ptr{NbpTreeBelief}
  diff_out::Ptr{NbpTreeBelief}
  down::Int32
  status::Int32
end

# ---- error mapping: status < 0 -> error() -> the clique Task fails -> monitorCSMs puts ERROR_STATUS on every
#      channel -> solveTree! throws CompositeException (CliqStateMachineUtils.jl:184-246, test/testCSMMonitor.jl:51)
function chk(rc::Integer)
  rc >= 0 && return Int(rc)
  return error("libnbp status $rc: " * unsafe_string(ccall((:nbp_last_error, libnbp), Cstring, ())))
end

# ---- contexts: pooled, one per concurrently running clique Task --------------------------------------------------
# A context owns its HIP stream and arena; creating one costs a few hipMallocs, so they are kept and reused.
mutable struct NbpCtx
  ptr::Ptr{Cvoid}
  N::Int
  nslots::Int
end

function NbpCtx(N::Int, nslots::Int; device::Int = 0, side::Int = 0)
  r = Ref{Ptr{Cvoid}}(C_NULL)
  chk(ccall((:nbp_ctx_create, libnbp), Int32,
            (Int32, Int32, Int32, Ptr{Cvoid}, Int64, Int32, Ref{Ptr{Cvoid}}),
            device, N, nslots, C_NULL, 0, side, r))
  c = NbpCtx(r[], N, nslots)
  finalizer(x -> ccall((:nbp_ctx_destroy, libnbp), Int32, (Ptr{Cvoid},), x.ptr), c)
  return c
end

const _POOL = Dict{Int, Vector{NbpCtx}}()      # N => idle contexts
const _POOL_LOCK = ReentrantLock()
const _POOL_SLOTS = 512                         # slots per pooled context (a clique needs nvars + nmsgs + densities + maxF * nvars: nbp_clique_slots)

"borrow a context for N particles and at least `nslots` slots; give it back with `release!`"
function acquire(N::Int, nslots::Int)
  lock(_POOL_LOCK) do
    idle = get!(_POOL, N, NbpCtx[])
    i = findfirst(c -> c.nslots >= nslots, idle)
    i === nothing || return splice!(idle, i)
    return NbpCtx(N, max(nslots, _POOL_SLOTS); side = 2N)
  end
end
release!(c::NbpCtx) = lock(() -> push!(get!(_POOL, c.N, NbpCtx[]), c), _POOL_LOCK)

function withctx(f, N::Int, nslots::Int)
  c = acquire(N, nslots)
  try
    return f(c)
  finally
    release!(c)
  end
end

# ---- manifolds and point layouts (include/nbp.h, "Host point layout") -------------------------------------------
manifoldcode(::Position{1}) = NBP_EUCLID1          # ContinuousScalar = Position{1}
manifoldcode(::Position{2}) = NBP_EUCLID2
manifoldcode(::Position{3}) = NBP_EUCLID3
manifoldcode(::Circular) = NBP_CIRCULAR
manifoldcode(vt::InferenceVariable) =
  getManifold(vt) isa typeof(SpecialEuclidean(2; vectors = HybridTangentRepresentation())) ? NBP_SE2 :
  error("libnbp: unsupported variable type $(typeof(vt))")
# the variable types a slot holds (tangent dimension <= NBP_MAXD = 3): a clique with any other -- ContinuousEuclid{4}, a user
# manifold -- takes the reference's method (upGibbsCliqueDensity / solveCliqDownFrontalProducts! below, `varsok`)
varok(::Union{Position{1}, Position{2}, Position{3}, Circular}) = true
varok(vt::InferenceVariable) = getManifold(vt) isa typeof(SpecialEuclidean(2; vectors = HybridTangentRepresentation()))
varsok(dfg::AbstractDFG, factors) = all(fc -> all(v -> varok(getVariableType(dfg, v)), getVariableOrder(fc)), factors)

pointdoubles(code::Int32) = code == NBP_SE2 ? 6 : (code == NBP_CIRCULAR ? 1 : Int(code))
tangentdim(code::Int32) = code == NBP_SE2 ? 3 : (code == NBP_CIRCULAR ? 1 : Int(code))

"points of a belief -> packed AoS Float64 (N x P)"
function packpoints(code::Int32, pts::AbstractVector)::Vector{Float64}
  P = pointdoubles(code)
  buf = Vector{Float64}(undef, P * length(pts))
  for (n, p) in enumerate(pts)
    o = (n - 1) * P
    if code == NBP_SE2                       # ArrayPartition(t::SVector{2}, R::SMatrix{2,2}): x, y, R11, R21, R12, R22
      t, R = p.x[1], p.x[2]
      buf[o + 1] = t[1]; buf[o + 2] = t[2]
      buf[o + 3] = R[1, 1]; buf[o + 4] = R[2, 1]; buf[o + 5] = R[1, 2]; buf[o + 6] = R[2, 2]
    else
      for d in 1:P
        buf[o + d] = p[d]
      end
    end
  end
  return buf
end

"packed AoS Float64 -> the point type the variable stores"
function wrappoints(vt::InferenceVariable, code::Int32, buf::Vector{Float64}, N::Int)
  P = pointdoubles(code)
  if code == NBP_SE2
    return [ArrayPartition(SVector{2, Float64}(buf[o + 1], buf[o + 2]),
                           SMatrix{2, 2, Float64}(buf[o + 3], buf[o + 4], buf[o + 5], buf[o + 6])) for o in 0:P:(P * (N - 1))]
  elseif code == NBP_CIRCULAR
    return [Float64[buf[n]] for n in 1:N]                       # Circular stores Vector{Vector{Float64}}
  else
    return [SVector{P, Float64}(ntuple(d -> buf[(n - 1) * P + d], P)) for n in 1:N]
  end
end

# ---- factors: the closed set libnbp implements (SURVEY a10) ---------------------------------------------------------
const NbpRelative = Union{LinearRelative, CircularCircular, EuclidDistance, ManifoldFactor}
const NbpPrior = Union{Prior, PriorCircular, ManifoldPrior}
const NbpUser = Union{NbpRelative, NbpPrior, Mixture, PartialPriorPassThrough}

factorkind(::Union{Prior, PriorCircular, ManifoldPrior}) = NBP_F_PRIOR
factorkind(::LinearRelative) = NBP_F_LINREL
factorkind(::CircularCircular) = NBP_F_CIRCULAR
factorkind(::ManifoldFactor) = NBP_F_SE2
factorkind(::EuclidDistance) = NBP_F_EUCLIDDIST
factorkind(m::Mixture) = factorkind(m.mechanics)
factorkind(::PartialPriorPassThrough) = NBP_F_PASSTHROUGH   # its density travels in nbp_clique_desc.factor_density

"the measurement of a differential factor of a joint upward message (useMsgLikelihoods): `_sft(newBel)` with newBel a
 ManifoldKernelDensity on the factor's manifold (addLikelihoodsDifferentialCHILD!, TreeMessageUtils.jl:279-335); nothing
 for every factor with a parametric measurement"
measkde(fnc) = (fnc isa Union{LinearRelative, CircularCircular, ManifoldFactor} && fnc.Z isa ManifoldKernelDensity) ? fnc.Z : nothing
measkde(::Union{Mixture, PartialPriorPassThrough}) = nothing

"one measurement-model component: weight, mean[3], lower Cholesky factor L[3][3] row-major (NBP_COMP_STRIDE doubles)"
function component(w::Real, Z)::Vector{Float64}
  row = zeros(Float64, NBP_COMP_STRIDE)
  row[1] = w
  if Z isa Uniform                                      # enum nbp_dist (include/nbp.h): scalar families ride in the last slot
    row[2] = minimum(Z); row[5] = maximum(Z) - minimum(Z); row[NBP_COMP_STRIDE] = 1.0
    return row
  elseif Z isa Rayleigh
    row[5] = scale(Z); row[NBP_COMP_STRIDE] = 2.0
    return row
  elseif Z isa AliasingScalarSampler                    # NBP_DIST_TABLE: the table travels in factor_density[f] (tablebuf)
    row[NBP_COMP_STRIDE] = 3.0
    return row
  end
  mu = Z isa Normal ? [mean(Z)] : collect(mean(Z))
  L = Z isa Normal ? fill(std(Z), 1, 1) : Matrix(cholesky(Symmetric(Matrix(cov(Z)))).L)
  for i in 1:min(3, length(mu))
    row[1 + i] = mu[i]
  end
  for i in 1:min(3, size(L, 1)), j in 1:i
    row[4 + 3 * (i - 1) + j] = L[i, j]
  end
  return row
end

function components(fnc)::Vector{Float64}
  if fnc isa Mixture                                   # Factors/Mixture.jl: components + diversity (Categorical)
    zs = collect(values(fnc.components))
    ws = probs(fnc.diversity)
    length(zs) <= NBP_MAXC || error("libnbp: at most $NBP_MAXC mixture components")
    flat = reduce(vcat, [component(ws[i], zs[i]) for i in eachindex(zs)])
    return vcat(flat, zeros(Float64, NBP_MAXC * NBP_COMP_STRIDE - length(flat)))
  end
  fnc isa PartialPriorPassThrough && return vcat([1.0], zeros(Float64, NBP_MAXC * NBP_COMP_STRIDE - 1))  # no measurement model
  if measkde(fnc) !== nothing                          # LinearRelative(::MKD) & co.: the measurement is the KDE in factor_meas_kde
    row = zeros(Float64, NBP_COMP_STRIDE)
    row[1] = 1.0; row[5] = 1.0; row[9] = 1.0; row[13] = 1.0      # weight, identity square-root covariance (unused)
    return vcat(row, zeros(Float64, (NBP_MAXC - 1) * NBP_COMP_STRIDE))
  end
  return vcat(component(1.0, fnc.Z), zeros(Float64, (NBP_MAXC - 1) * NBP_COMP_STRIDE))
end
ncomponents(fnc) = fnc isa Mixture ? length(fnc.components) : 1

"the AliasingScalarSampler among a factor's measurement models (entities/AliasScalarSampling.jl:13), or nothing"
function tablesampler(fnc)
  fnc isa PartialPriorPassThrough && return nothing
  zs = fnc isa Mixture ? collect(values(fnc.components)) : (hasfield(typeof(fnc), :Z) ? [fnc.Z] : [])
  i = findfirst(z -> z isa AliasingScalarSampler, zs)
  return i === nothing ? nothing : zs[i]
end
"the sampler's table as libnbp holds it: a belief on Euclid(2), row 0 the domain, row 1 the cumulative weights (enum nbp_dist)"
function tablebuf(z::AliasingScalarSampler)
  cum = cumsum(collect(Float64, z.weights))
  cum[end] = 1.0
  return BeliefBuf(NBP_EUCLID2, [[z.domain[i], cum[i]] for i in eachindex(cum)], ones(Float64, 2), zeros(Float64, 2), length(cum))
end

partialmask(fnc) = hasfield(typeof(fnc), :partial) ? Int32(sum(1 << (k - 1) for k in fnc.partial)) : Int32(0)

pad(v::AbstractVector{T}, n::Int, z::T) where {T} = ntuple(i -> i <= length(v) ? v[i] : z, n)

"nbp_factor_spec of a DFG factor; `index`: variable label -> 0-based position in the clique's variable list"
function factorspec(fct::DFGFactor, index::Dict{Symbol, Int})::NbpFactorSpec
  ccw = _getCCW(fct)
  fnc = getFactorType(fct)
  vars = Int32[index[v] for v in getVariableOrder(fct)]
  mh = ccw.hyporecipe.hypotheses                         # parsed Categorical, certain variables carry 0.0 (FactorGraph.jl:639-651)
  return NbpFactorSpec(factorkind(fnc), Int32(length(vars)), pad(vars, NBP_MAXV, Int32(0)), Int32(ncomponents(fnc)),
                       Int32(mh === nothing ? 0 : 1), partialmask(fnc),
                       pad(mh === nothing ? Float64[] : collect(Float64, mh.p), NBP_MAXV, 0.0),
                       Float64(ccw.nullhypo), Float64(ccw.inflation), Tuple(components(fnc)))
end

# `iters`: the Gibbs iterations of the iterated variables -- upGibbsCliqueDensity's own argument (SolveTree.jl:171,216-227),
# MCIters of the down solve; SolverParams.gibbsIters where the caller has neither
solverparams(sp, N::Int, iters::Int = sp.gibbsIters) = NbpSolverParams(Int32(N), Int32(iters), Int32(sp.inflateCycles), Int32(1),
                                           Int32(sp.upsolve), Int32(sp.downsolve), Int32(sp.limitfixeddown),
                                           sp.alwaysFreshMeasurements ? Int32(0) : NBP_SOLVER_STORED_MEASUREMENTS,
                                           Float64(sp.spreadNH), Float64(sp.inflation), Float64(sp.nullSurplusAdd))

# the measurement models libnbp samples (enum nbp_dist + MvNormal): anything else -- the reference accepts every samplable
# belief (ManifoldSampling.jl:121-145) -- sends its clique down the generic path, factor by factor (INTEGRATION.md 1)
const NbpMeas = Union{Normal, MvNormal, Uniform, Rayleigh, AliasingScalarSampler}
measok(fnc::Mixture) = all(z -> z isa NbpMeas, values(fnc.components)) && measok(fnc.mechanics, true)
measok(::Union{PartialPriorPassThrough, MsgPrior}) = true
measok(fnc, mechanicsonly::Bool = false) = mechanicsonly || (hasfield(typeof(fnc), :Z) && fnc.Z isa Union{NbpMeas, ManifoldKernelDensity})
supported(fct::DFGFactor) = getFactorType(fct) isa Union{NbpUser, MsgPrior{<:ManifoldKernelDensity}} && measok(getFactorType(fct))
# ... for a solve with N particles: the table of an AliasingScalarSampler lives in a belief slot of N rows, a longer one is
# refused by nbp_clique_* (NBP_ERR_RANGE) -- such a factor's clique takes the generic path instead
function supported(fct::DFGFactor, N::Int)
  supported(fct) || return false
  tb = tablesampler(getFactorType(fct))
  return tb === nothing || length(tb.domain) <= N
end

# ---- beliefs at the boundary ------------------------------------------------------------------------------------------
"host buffers of one TreeBelief; keeps them alive next to the C view"
struct BeliefBuf
  pts::Vector{Float64}
  bw::Vector{Float64}
  ipc::Vector{Float64}
  n::Int
end
function BeliefBuf(code::Int32, pts::AbstractVector, bw::AbstractVector, ipc::AbstractVector, N::Int)
  D = tangentdim(code)
  buf = packpoints(code, pts)
  resize!(buf, max(length(buf), N * pointdoubles(code)))          # room for the N points that come back
  ipc_ = length(ipc) == D ? collect(Float64, ipc) : zeros(Float64, D)
  return BeliefBuf(buf, collect(Float64, bw[1:D]), ipc_, length(pts))
end
cview(b::BeliefBuf) = NbpTreeBelief(pointer(b.pts), pointer(b.bw), pointer(b.ipc), Int32(b.n), Int32(0))

function BeliefBuf(dfg::AbstractDFG, sym::Symbol, solveKey::Symbol, N::Int)
  vnd = getSolverData(getVariable(dfg, sym), solveKey)
  code = manifoldcode(getVariableType(dfg, sym))
  return BeliefBuf(code, vnd.val, vnd.bw[:, 1], vnd.infoPerCoord, N)
end
function BeliefBuf(mkd::ManifoldKernelDensity, code::Int32, ipc, N::Int)
  return BeliefBuf(code, getPoints(mkd, false), getBW(mkd)[:, 1], ipc, N)
end
"the density of a PartialPriorPassThrough (calcProposalBelief, ApproxConv.jl:196-227) in the variable's point layout:
 its partial coordinates filled in, the others at the identity (antimarginal), bandwidth zero where it says nothing"
function BeliefBuf(fnc::PartialPriorPassThrough, vt::InferenceVariable, code::Int32)
  mkd = fnc.Z.heatmap.densityFnc
  pts = getPoints(mkd, false)
  D, P = tangentdim(code), pointdoubles(code)
  part = Int[fnc.partial...]
  buf = zeros(Float64, length(pts) * P)
  bw = zeros(Float64, D)
  h = getBW(mkd)[:, 1]
  for (i, k) in enumerate(part)
    bw[k] = h[i]
  end
  for (n, p) in enumerate(pts)
    c = zeros(Float64, D)
    for (i, k) in enumerate(part)
      c[k] = p[i]
    end
    o = (n - 1) * P
    if code == NBP_SE2
      buf[o + 1] = c[1]; buf[o + 2] = c[2]
      buf[o + 3] = cos(c[3]); buf[o + 4] = sin(c[3]); buf[o + 5] = -sin(c[3]); buf[o + 6] = cos(c[3])
    else
      buf[(o + 1):(o + P)] .= c
    end
  end
  return BeliefBuf(buf, bw, ones(Float64, D), length(pts))
end
"the measurement KDE of a differential factor as nbp_clique_desc.factor_meas_kde takes it: the tangent coordinates of its
 points at the identity of the factor's manifold (zdim doubles per point, packed like an Euclid(zdim) belief) + bandwidth"
function BeliefBuf(fnc, mkd::ManifoldKernelDensity, N::Int)
  M = getManifold(fnc)
  e0 = getPointIdentity(M)
  coords = [collect(Float64, vee(M, e0, log(M, e0, p))) for p in getPoints(mkd, false)]
  zd = length(coords[1])
  return BeliefBuf(Int32(zd), coords, getBW(mkd)[:, 1], zeros(Float64, zd), N)    # manifold code of Euclid(zd) = zd
end
const _NOBELIEF = NbpTreeBelief(Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), Int32(0), Int32(0))

# ---- joint upward messages, sending side (useMsgLikelihoods) ----------------------------------------------------------------
# WHICH separator pairs get a differential factor is symbolic and stays the reference's: the selection of
# addLikelihoodsDifferentialCHILD! (TreeMessageUtils.jl:296-317: separators by decreasing dimension, every later one as
# partner, the path between them homogeneous and of the default relative type).  The numeric half of that function --
# approxDeconv of the dummy factor between the solved beliefs and manikde! of the predicted measurements -- is what
# nbp_clique_upsolve_joint does on the device, in the same call as the up solve (diff_a / diff_b / diff_kind -> diff_out).
"pairs (0-based indices into the clique's variable list), factor kinds, default factor types and the buffers of their KDEs"
struct DiffPlan
  a::Vector{Int32}
  b::Vector{Int32}
  kind::Vector{Int32}
  syms::Vector{Tuple{Symbol, Symbol}}
  ftype::Vector{Any}                           # selectFactorType(...) of the pair: LinearRelative{N}, CircularCircular, ...
  bufs::Vector{BeliefBuf}                      # zdim doubles per point, N points, + bandwidth: what diff_out[i] points into
  out::Vector{NbpTreeBelief}
end
DiffPlan() = DiffPlan(Int32[], Int32[], Int32[], Tuple{Symbol, Symbol}[], Any[], BeliefBuf[], NbpTreeBelief[])

function diffplan(dfg::AbstractDFG, seps::Vector{Symbol}, index::Dict{Symbol, Int}, N::Int)
  plan = DiffPlan()
  (getSolverParams(dfg).useMsgLikelihoods && length(seps) > 1) || return plan
  per = sortperm(getDimension.(getVariable.(dfg, seps)); rev = true)
  dec = seps[per]
  acc = reverse(dec)
  already = Symbol[]
  for s1 in dec
    push!(already, s1)
    for s2 in setdiff(acc, already)
      isHom, ftyps = isPathFactorsHomogeneous(dfg, s1, s2)
      isHom || continue
      _sft = selectFactorType(dfg, s1, s2)
      sft = _sft()
      (typeof(sft).name == ftyps[1] && sft isa Union{LinearRelative, CircularCircular, ManifoldFactor}) || continue
      zd = manifold_dimension(getManifold(sft))
      push!(plan.a, Int32(index[s1])); push!(plan.b, Int32(index[s2])); push!(plan.kind, factorkind(sft))
      push!(plan.syms, (s1, s2)); push!(plan.ftype, _sft)
      push!(plan.bufs, BeliefBuf(Int32(zd), [zeros(Float64, zd) for _ in 1:N], ones(Float64, zd), zeros(Float64, zd), N))
    end
  end
  append!(plan.out, cview.(plan.bufs))
  return plan
end

"the KDEs nbp_clique_upsolve_joint left for the clique sub graph `subfg`, until prepCliqueMsgUp asks for them"
const _DIFFS = Dict{UInt, Vector{NamedTuple}}()
const _DIFFS_LOCK = ReentrantLock()

"`_sft(newBel)` per pair from what the device wrote: tangent coordinates at the identity -> points of the factor's manifold"
function stashdiffs!(subfg::AbstractDFG, plan::DiffPlan, N::Int)
  ret = NamedTuple[]
  for (i, (s1, s2)) in enumerate(plan.syms)
    _sft = plan.ftype[i]
    M = getManifold(_sft())
    e0 = getPointIdentity(M)
    b = plan.bufs[i]
    zd = length(b.bw)
    pts = [exp(M, e0, hat(M, e0, b.pts[((n - 1) * zd + 1):(n * zd)])) for n in 1:N]
    push!(ret, (; variables = [s1; s2], likelihood = _sft(manikde!(M, pts; bw = b.bw))))
  end
  lock(_DIFFS_LOCK) do
    _DIFFS[objectid(subfg)] = ret
  end
  return nothing
end

"""
    addLikelihoodsDifferentialCHILD!(cliqSubFG, seps, tfg; solveKey)

TreeMessageUtils.jl:279-335 for a clique whose up solve ran on the device: the differential factors were computed by
nbp_clique_upsolve_joint in that call (same pairs, same order) and are handed over here; a clique that was solved by the
generic path (unsupported factor) takes the generic method.
"""
function addLikelihoodsDifferentialCHILD!(cliqSubFG::GraphsDFG, seps::Vector{Symbol}, tfg::AbstractDFG; solveKey::Symbol = :default)
  ret = lock(_DIFFS_LOCK) do
    pop!(_DIFFS, objectid(cliqSubFG), nothing)
  end
  ret === nothing && return invoke(addLikelihoodsDifferentialCHILD!, Tuple{AbstractDFG, Vector{Symbol}, AbstractDFG}, cliqSubFG, seps, tfg; solveKey)
  out = IncrementalInference.MsgRelativeType()
  foreach(r -> push!(out, r), ret)
  return out
end

# ---- the clique seam --------------------------------------------------------------------------------------------------
"everything one nbp_clique_* call needs, with the Julia arrays the C struct points into"
struct CliquePack
  labels::Vector{Symbol}
  codes::Vector{Int32}
  margin::Vector{Int32}
  specs::Vector{NbpFactorSpec}
  lists::NTuple{4, Vector{Int32}}
  msgvar::Vector{Int32}
  msgbuf::Vector{BeliefBuf}
  msgs::Vector{NbpTreeBelief}
  bufs::Vector{BeliefBuf}
  beliefs::Vector{NbpTreeBelief}
  densbuf::Vector{Union{Nothing, BeliefBuf}}   # per user factor: the density of a PartialPriorPassThrough
  dens::Vector{NbpTreeBelief}
  kdebuf::Vector{Union{Nothing, BeliefBuf}}    # per user factor: the measurement KDE of a differential factor (joint messages)
  kdes::Vector{NbpTreeBelief}
  diff::DiffPlan                               # the differential factors this clique sends up (joint messages, sending side)
end

function packclique(dfg::AbstractDFG, cliq::TreeClique, solveKey::Symbol, N::Int, labels::Vector{Symbol}, factors::Vector{<:DFGFactor};
                    senddiffs::Bool = false)
  cd = getCliqueData(cliq)
  index = Dict{Symbol, Int}(l => i - 1 for (i, l) in enumerate(labels))
  all(supported, factors) || error("libnbp: unsupported factor type in clique $(cliq.id)")
  user = [f for f in factors if !(getFactorType(f) isa MsgPrior)]
  msgf = [f for f in factors if getFactorType(f) isa MsgPrior]
  codes = Int32[manifoldcode(getVariableType(dfg, l)) for l in labels]
  margin = Int32[getSolverData(getVariable(dfg, l), solveKey).ismargin ? 1 : 0 for l in labels]
  specs = NbpFactorSpec[factorspec(f, index) for f in user]
  ids(v) = Int32[index[s] for s in v if haskey(index, s)]
  lists = (ids(cd.directFrtlMsgIDs), ids(cd.msgskipIDs), ids(cd.itervarIDs), ids(cd.directPriorMsgIDs))
  msgvar = Int32[index[getVariableOrder(f)[1]] for f in msgf]
  msgbuf = BeliefBuf[BeliefBuf(getFactorType(f).Z, codes[index[getVariableOrder(f)[1]] + 1], getFactorType(f).infoPerCoord, N) for f in msgf]
  bufs = BeliefBuf[BeliefBuf(dfg, l, solveKey, N) for l in labels]
  densbuf = Union{Nothing, BeliefBuf}[
    (fnc = getFactorType(f); v = getVariableOrder(f)[1];
     fnc isa PartialPriorPassThrough ? BeliefBuf(fnc, getVariableType(dfg, v), codes[index[v] + 1]) :
     (tablesampler(fnc) === nothing ? nothing : tablebuf(tablesampler(fnc)))) for f in user]
  dens = NbpTreeBelief[b === nothing ? _NOBELIEF : cview(b) for b in densbuf]
  kdebuf = Union{Nothing, BeliefBuf}[
    (fnc = getFactorType(f); z = measkde(fnc); z === nothing ? nothing : BeliefBuf(fnc, z, N)) for f in user]
  kdes = NbpTreeBelief[b === nothing ? _NOBELIEF : cview(b) for b in kdebuf]
  return CliquePack(labels, codes, margin, specs, lists, msgvar, msgbuf, cview.(msgbuf), bufs, cview.(bufs), densbuf, dens, kdebuf, kdes,
                    senddiffs ? diffplan(dfg, Symbol[cd.separatorIDs...], index, N) : DiffPlan())
end

ptr_or_null(v::Vector{T}) where {T} = isempty(v) ? Ptr{T}(C_NULL) : pointer(v)

function cliquedesc(cliq::TreeClique, p::CliquePack, nfrontals::Int, nseparators::Int)
  return NbpCliqueDesc(Int32(cliq.id.value), Int32(length(p.labels)), Int32(nfrontals), Int32(nseparators),
                       pointer(p.codes), pointer(p.margin), Int32(length(p.specs)), ptr_or_null(p.specs),
                       Int32(length(p.lists[1])), Int32(length(p.lists[2])), Int32(length(p.lists[3])), Int32(length(p.lists[4])),
                       ptr_or_null(p.lists[1]), ptr_or_null(p.lists[2]), ptr_or_null(p.lists[3]), ptr_or_null(p.lists[4]),
                       Int32(length(p.msgvar)), ptr_or_null(p.msgvar), ptr_or_null(p.msgs),
                       any(b -> b !== nothing, p.densbuf) ? pointer(p.dens) : Ptr{NbpTreeBelief}(C_NULL),
                       # joint messages (useMsgLikelihoods), receiving side: the differential factors of a child's message are
                       # entries of `factors` whose measurement is a KDE; sending side: the pairs of diffplan (up solve only)
                       any(b -> b !== nothing, p.kdebuf) ? pointer(p.kdes) : Ptr{NbpTreeBelief}(C_NULL),
                       Int32(length(p.diff.a)), Int32(0), ptr_or_null(p.diff.a), ptr_or_null(p.diff.b), ptr_or_null(p.diff.kind))
end

"write the beliefs libnbp returned back into the sub graph: setValKDE!(vnd, pts, bw, setinit, ipc) (FactorGraph.jl:250-297)"
function unpack!(dfg::AbstractDFG, p::CliquePack, solveKey::Symbol, N::Int, which)
  for (i, l) in enumerate(p.labels)
    l in which || continue
    vt = getVariableType(dfg, l)
    b = p.bufs[i]
    # the count libnbp wrote back: N for a solved belief, the density's own count when the variable's only factor is a
    # PartialPriorPassThrough ("PassThrough transfers the full point count to the graph", testSpecialEuclidean2Mani.jl:394)
    n = Int(p.beliefs[i].n_pts)
    (1 <= n <= N) || error("libnbp: belief $(l) came back with $(n) points")
    setValKDE!(getSolverData(getVariable(dfg, l), solveKey), wrappoints(vt, p.codes[i], b.pts, n), reshape(b.bw, :, 1), true, b.ipc)
  end
  return nothing
end

# ---- gathering the clique calls that are ready (opt-in: IIFNbpExt.BATCH_CLIQUES[] = true) ------------------------------------
# The state machines run as one task per clique and reach their solve step independently; cliques of one tree level do not
# depend on each other.  With BATCH_CLIQUES the calls are queued, and a dispatcher task hands everything that is waiting to
# ONE nbp_clique_solve_batch: one transfer of beliefs each way and shared launches (DESIGN.md 6: 926 ms -> 41 ms per solve
# of the config-2 graph when whole levels arrive together).  The state machines stay as they are: each waits for its own
# result.  Same posteriors either way (the random streams are keyed by clique, not by batch).
# Round 5: the batch goes through nbp_clique_submit_batch / nbp_clique_wait (nbp_host.h) -- the call returns once everything
# is queued, so the dispatcher overlaps its own planning of the next batch with the device's work on this one.  Resident
# beliefs (NbpTreeBelief.handle: a message that stays on the device between the child's call and the parent's) are offered
# by the C ABI and exercised by examples/solve_by_clique_calls.c (47 -> 32 ms per solve of the config-2 graph); using them
# from the state machines means a TreeBelief whose `val` is fetched on first read, which this shim does not attempt.
const BATCH_CLIQUES = Ref(false)
struct PendingClique
  N::Int
  need::Int
  sp::Base.RefValue{NbpSolverParams}
  q::Base.RefValue{NbpCliqueDesc}
  seed::UInt64
  p::CliquePack
  down::Bool
  done::Channel{Any}   # the status (Int32) or the exception
end
const _PENDING = Channel{PendingClique}(Inf)
const _DISPATCHER = Ref{Union{Nothing, Task}}(nothing)

function dispatchcliques()
  while true
    batch = PendingClique[take!(_PENDING)]
    yield()                                         # the other clique tasks that are ready get to queue up
    while isready(_PENDING)
      push!(batch, take!(_PENDING))
    end
    for N in unique(b.N for b in batch)             # one context per particle count
      group = [b for b in batch if b.N == N]
      try
        reqs = NbpCliqueRequest[NbpCliqueRequest(Base.unsafe_convert(Ptr{NbpSolverParams}, b.sp), Base.unsafe_convert(Ptr{NbpCliqueDesc}, b.q),
                                                 b.seed, pointer(b.p.beliefs), ptr_or_null(b.p.diff.out), b.down ? 1 : 0, 0) for b in group]
        # submit, then wait: the batch is queued on the library stream (nbp_clique_submit_batch returns without waiting for
        # the device), so with more than one Julia thread the wait runs elsewhere and this dispatcher is free to gather, plan
        # and queue whatever became ready in the meantime behind it -- batches run in submission order.  The context stays
        # borrowed, and `group` / `reqs` stay referenced by the closure (Julia's collector does not move objects), until
        # nbp_clique_wait has unpacked the results.
        ctx = acquire(N, sum(b.need for b in group))
        ticket = Ref{Ptr{Cvoid}}(C_NULL)
        try
          GC.@preserve group reqs chk(ccall((:nbp_clique_submit_batch, libnbp), Int32, (Ptr{Cvoid}, Ptr{NbpCliqueRequest}, Int32, Ref{Ptr{Cvoid}}),
                                            ctx.ptr, reqs, length(reqs), ticket))
        catch
          release!(ctx)
          rethrow()
        end
        finish = () -> begin
          try
            GC.@preserve group reqs chk(ccall((:nbp_clique_wait, libnbp), Int32, (Ptr{Cvoid},), ticket[]))
            foreach((b, r) -> put!(b.done, r.status), group, reqs)
          catch err
            foreach(b -> put!(b.done, err), group)     # (a hard error of the device: every clique of the batch sees it)
          finally
            release!(ctx)
          end
        end
        Threads.nthreads() > 1 ? errormonitor(Threads.@spawn finish()) : finish()
      catch
        # one request of the batch is at fault (unsupported input, slot overflow): the others must not fail with it -- each
        # request again on its own, so that only the offending clique task sees the error (monitorCSMs takes it from there).
        # Which calls share a batch depends on task timing; the results do not (the random streams are keyed per clique).
        for b in group
          try
            req = NbpCliqueRequest[NbpCliqueRequest(Base.unsafe_convert(Ptr{NbpSolverParams}, b.sp), Base.unsafe_convert(Ptr{NbpCliqueDesc}, b.q),
                                                    b.seed, pointer(b.p.beliefs), ptr_or_null(b.p.diff.out), b.down ? 1 : 0, 0)]
            GC.@preserve b req withctx(N, b.need) do ctx
              chk(ccall((:nbp_clique_solve_batch, libnbp), Int32, (Ptr{Cvoid}, Ptr{NbpCliqueRequest}, Int32), ctx.ptr, req, 1))
            end
            put!(b.done, req[1].status)
          catch err
            put!(b.done, err)
          end
        end
      end
    end
  end
end

function runclique_batched(q::NbpCliqueDesc, sp::NbpSolverParams, N::Int, need::Int, seed::UInt64, p::CliquePack, down::Bool)
  lock(_POOL_LOCK) do
    (_DISPATCHER[] === nothing || istaskdone(_DISPATCHER[])) && (_DISPATCHER[] = errormonitor(@async dispatchcliques()))
  end
  pc = PendingClique(N, need, Ref(sp), Ref(q), seed, p, down, Channel{Any}(1))
  put!(_PENDING, pc)
  r = take!(pc.done)
  r isa Exception && throw(r)
  return r::Int32
end

function runclique(sym::Symbol, dfg::AbstractDFG, cliq::TreeClique, solveKey::Symbol, N::Int, p::CliquePack, nfr::Int, nsep::Int, seed::UInt64,
                   iters::Int = getSolverParams(dfg).gibbsIters)
  q = cliquedesc(cliq, p, nfr, nsep)
  sp = solverparams(getSolverParams(dfg), N, iters)
  status = Ref{Int32}(0)
  if BATCH_CLIQUES[]
    need = GC.@preserve p chk(ccall((:nbp_clique_slots, libnbp), Int32, (Ref{NbpCliqueDesc},), q))
    return runclique_batched(q, sp, N, Int(need), seed, p, sym !== :up)
  end
  GC.@preserve p begin
    need = chk(ccall((:nbp_clique_slots, libnbp), Int32, (Ref{NbpCliqueDesc},), q))
    withctx(N, need) do ctx
      if sym === :up && !isempty(p.diff.a)      # the up solve and the clique's differential factors in one call
        chk(ccall((:nbp_clique_upsolve_joint, libnbp), Int32,
                  (Ptr{Cvoid}, Ref{NbpSolverParams}, Ref{NbpCliqueDesc}, UInt64, Ptr{NbpTreeBelief}, Ptr{NbpTreeBelief}, Ref{Int32}),
                  ctx.ptr, sp, q, seed, p.beliefs, p.diff.out, status))
      elseif sym === :up
        chk(ccall((:nbp_clique_upsolve, libnbp), Int32,
                  (Ptr{Cvoid}, Ref{NbpSolverParams}, Ref{NbpCliqueDesc}, UInt64, Ptr{NbpTreeBelief}, Ref{Int32}),
                  ctx.ptr, sp, q, seed, p.beliefs, status))
      else
        chk(ccall((:nbp_clique_downsolve, libnbp), Int32,
                  (Ptr{Cvoid}, Ref{NbpSolverParams}, Ref{NbpCliqueDesc}, UInt64, Ptr{NbpTreeBelief}, Ref{Int32}),
                  ctx.ptr, sp, q, seed, p.beliefs, status))
      end
    end
  end
  return status[]
end

"""
    upGibbsCliqueDensity(dfg, cliq, solveKey, inmsgs, N, dbg, iters, logger)

The clique up solve on the device: src/services/SolveTree.jl:164-239 (called from approxCliqMarginalUp!,
CliqStateMachineUtils.jl:375-385).  `dfg` is the clique sub graph with the children's messages already added as
MsgPrior factors (addMsgFactors!, TreeMessageUtils.jl:542-578); like the reference it is updated in place
(setBelief! of every variable the schedule touches) and the beliefs are returned as `Dict{Symbol,TreeBelief}`.
"""
function upGibbsCliqueDensity(dfg::GraphsDFG, cliq::TreeClique, solveKey::Symbol, inmsgs, N::Int = getSolverParams(dfg).N,
                              dbg::Bool = false, iters::Int = 3, logger = ConsoleLogger())
  cd = getCliqueData(cliq)
  labels = Symbol[cd.frontalIDs; cd.separatorIDs]
  factors = DFGFactor[getFactor(dfg, f) for f in lsf(dfg)]
  (all(f -> supported(f, N), factors) && varsok(dfg, factors)) || return invoke(upGibbsCliqueDensity, Tuple{AbstractDFG, TreeClique, Symbol, Any, Int, Bool, Int, Any},
                                           dfg, cliq, solveKey, inmsgs, N, dbg, iters, logger)      # generic CPU path
  p = packclique(dfg, cliq, solveKey, N, labels, factors; senddiffs = true)
  runclique(:up, dfg, cliq, solveKey, N, p, length(cd.frontalIDs), length(cd.separatorIDs), rand(UInt64), iters)
  isempty(p.diff.a) || stashdiffs!(dfg, p.diff, N)     # handed to prepCliqueMsgUp through addLikelihoodsDifferentialCHILD!
  # what the four fmcmc! calls of the reference touch (SolveTree.jl:193-235): marginalized variables are skipped inside
  # fmcmc! (:61) but still reported by compileFMCMessages (:32-45), so the lists decide, not the margin flags
  touched = Symbol[l for l in labels if l in cd.directFrtlMsgIDs || l in cd.msgskipIDs || l in cd.itervarIDs || l in cd.directPriorMsgIDs]
  unpack!(dfg, p, solveKey, N, [l for (i, l) in enumerate(labels) if l in touched && p.margin[i] == 0])
  d = Dict{Symbol, TreeBelief}()
  for l in touched                                           # compileFMCMessages, SolveTree.jl:32-45
    d[l] = TreeBelief(getVariable(dfg, l), solveKey)
  end
  return d
end

"""
    solveCliqDownFrontalProducts!(subfg, cliq, opts, logger; solveKey, MCIters)

The clique down solve on the device (CliqStateMachineUtils.jl:479-571): `subfg` already holds the parent's separator
values (updateSubFgFromDownMsgs!) and every factor of the frontals (addDownVariableFactors!, CliqueStateMachine.jl:823-835).
"""
function solveCliqDownFrontalProducts!(subfg::GraphsDFG, cliq::TreeClique, opts::SolverParams, logger = ConsoleLogger();
                                       solveKey::Symbol = :default, MCIters::Int = 3)
  cd = getCliqueData(cliq)
  inclq = Symbol[cd.frontalIDs; cd.separatorIDs]
  factors = DFGFactor[]
  for v in cd.frontalIDs, f in ls(subfg, v)
    fc = getFactor(subfg, f)
    (getFactorType(fc) isa MsgPrior || fc in factors) || push!(factors, fc)
  end
  (MCIters == 3 && all(f -> supported(f, opts.N), factors) && varsok(subfg, factors)) || return invoke(solveCliqDownFrontalProducts!, Tuple{AbstractDFG, TreeClique, SolverParams, Any},
                                                           subfg, cliq, opts, logger; solveKey, MCIters)
  others = Symbol[]
  for fc in factors, u in getVariableOrder(fc)
    (u in inclq || u in others) || push!(others, u)
  end
  labels = Symbol[inclq; others]
  N = opts.N
  p = packclique(subfg, cliq, solveKey, N, labels, factors)
  runclique(:down, subfg, cliq, solveKey, N, p, length(cd.frontalIDs), length(cd.separatorIDs), rand(UInt64))
  unpack!(subfg, p, solveKey, N, cd.frontalIDs)
  return nothing
end

# ---- the finer seams (host buffers, one ccall per reference function) ---------------------------------------------------
"AMP.manikde!(M, pts) bandwidth selection (ApproxConv.jl:38,41; FGOSUtils.jl:118-128)"
function nbpbandwidth(vt::InferenceVariable, pts::AbstractVector)
  code = manifoldcode(vt)
  buf = packpoints(code, pts)
  bw = Vector{Float64}(undef, tangentdim(code))
  withctx(length(pts), 4) do ctx
    GC.@preserve buf bw chk(ccall((:nbp_kde_bandwidth, libnbp), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}),
                                  ctx.ptr, code, buf, bw))
  end
  return bw
end

"AMP.manifoldProduct(dens, M; Niter, oldPoints, N) + rebandwidth (GraphProductOperations.jl:53-60)"
function nbpproduct(vt::InferenceVariable, dens::Vector{<:ManifoldKernelDensity}; Niter::Int = 1, oldPoints = nothing,
                    N::Int = Npts(dens[1]), partials::Vector{UInt8} = zeros(UInt8, length(dens)))
  code = manifoldcode(vt)
  F = length(dens)
  pts = [packpoints(code, getPoints(d, false)) for d in dens]
  bws = [collect(Float64, getBW(d)[:, 1]) for d in dens]
  old = oldPoints === nothing ? Float64[] : packpoints(code, oldPoints)
  out = Vector{Float64}(undef, N * pointdoubles(code))
  bw = Vector{Float64}(undef, tangentdim(code))
  lbl = Vector{Int32}(undef, N * F)
  withctx(N, F + 2) do ctx
    GC.@preserve pts bws old out bw lbl partials begin
      pp, pb = pointer.(pts), pointer.(bws)
      chk(ccall((:nbp_manifold_product, libnbp), Int32,
                (Ptr{Cvoid}, Int32, Int32, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{UInt8}, Ptr{Float64}, Int32, UInt64,
                 Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                ctx.ptr, code, F, pp, pb, partials, isempty(old) ? Ptr{Float64}(C_NULL) : pointer(old), Niter, rand(UInt64), out, bw, lbl))
    end
  end
  return manikde!(getManifold(vt), wrappoints(vt, code, out, N); bw), reshape(lbl, F, N)
end

"nbp_proposal_desc of approxConvBelief(dfg, fct, target): the slot fields stay 0 (nbp_conv fills them)"
function proposaldesc(dfg::AbstractDFG, fct::DFGFactor, target::Symbol; nullSurplus::Real = 0.0, seed::UInt64 = rand(UInt64))
  ccw = _getCCW(fct)
  fnc = getFactorType(fct)
  sp = getSolverParams(dfg)
  order = getVariableOrder(fct)
  mh = ccw.hyporecipe.hypotheses
  flags = Int32(0)
  if mh !== nothing          # bit 0: multihypo; bit 7: isinit flags present; bit 8+k: variable k initialised
    flags = Int32(1 | 0x80)
    for (k, v) in enumerate(order)
      isInitialized(dfg, v) && (flags |= Int32(1) << (7 + k))
    end
  end
  return NbpProposalDesc(factorkind(fnc), manifoldcode(getVariableType(dfg, target)), Int32(length(order)),
                         Int32(findfirst(==(target), order) - 1), ntuple(_ -> Int32(0), NBP_MAXV), Int32(0),
                         Int32(ncomponents(fnc)), flags, Int32(sp.inflateCycles), Int32(-1), Int32(-1), Int32(0),
                         partialmask(fnc), Int32(0), Int32(0),
                         pad(mh === nothing ? Float64[] : collect(Float64, mh.p), NBP_MAXV, 0.0),
                         Float64(max(ccw.nullhypo, nullSurplus)), Float64(ccw.inflation), Float64(sp.spreadNH),
                         Tuple(components(fnc)), seed, UInt64(0))
end

"factor seam: approxConvBelief(dfg, fct, target) (ApproxConv.jl:4-45) as one nbp_conv call"
function approxConvBelief(dfg::AbstractDFG, fct::DFGFactor{<:CommonConvWrapper{<:NbpUser}}, target::Symbol,
                          measurement::AbstractVector = Tuple[]; solveKey::Symbol = :default,
                          N::Int = getSolverParams(dfg).N, nullSurplus::Real = 0, skipSolve::Bool = false)
  order = getVariableOrder(fct)
  vt = getVariableType(dfg, target)
  code = manifoldcode(vt)
  bufs = [packpoints(manifoldcode(getVariableType(dfg, v)), getVal(getVariable(dfg, v); solveKey)) for v in order]
  d = proposaldesc(dfg, fct, target; nullSurplus)
  out = Vector{Float64}(undef, N * pointdoubles(code))
  bw = Vector{Float64}(undef, tangentdim(code))
  mh = Vector{Int32}(undef, N)
  withctx(N, length(order) + 1) do ctx
    GC.@preserve bufs out bw mh begin
      ptrs = pointer.(bufs)
      chk(ccall((:nbp_conv, libnbp), Int32,
                (Ptr{Cvoid}, Ref{NbpProposalDesc}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                ctx.ptr, d, ptrs, C_NULL, C_NULL, out, bw, mh))
    end
  end
  return manikde!(getManifold(vt), wrappoints(vt, code, out, N); bw)
end

end # module
