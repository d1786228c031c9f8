# Smoke tests of ext/IIFNbpExt.jl for a maintainer's machine (Julia >= 1.10, IncrementalInference v0.35, libnbp.so on
# LD_LIBRARY_PATH or ENV["LIBNBP"]).  NOT executed in the build container (no Julia there: INTEGRATION.md 1); what it checks
# is pinned structurally by tests/test_julia_shim_layout.py.
#
#   julia --project -e 'include("ext/test/runtests.jl")'
using Test
using IncrementalInference
include(joinpath(@__DIR__, "..", "IIFNbpExt.jl"))

@testset "the overrides do not replace the reference's methods" begin
  for (f, sig) in ((upGibbsCliqueDensity, Tuple{AbstractDFG, TreeClique, Symbol, Any, Int, Bool, Int, Any}),
                   (IncrementalInference.solveCliqDownFrontalProducts!, Tuple{AbstractDFG, TreeClique, SolverParams, Any}),
                   (addLikelihoodsDifferentialCHILD!, Tuple{AbstractDFG, Vector{Symbol}, AbstractDFG}))
    m = which(f, sig)
    @test parentmodule(m) === IncrementalInference          # the generic method is still the reference's
    @test length(methods(f)) >= 2
  end
end

@testset "fallback: a clique with nothing stashed takes the reference's differential factors" begin
  fg = generateGraph_LineStep(4; poseEvery = 1, landmarkEvery = 5, posePriorsAt = [0], sightDistance = 2, solverParams = SolverParams(; useMsgLikelihoods = true))
  initAll!(fg)
  # a direct call with no device solve in front: nothing is stashed for this graph, the generic method must answer (the
  # round-4 file recursed here until StackOverflowError)
  ret = addLikelihoodsDifferentialCHILD!(fg, [:x0, :x1])
  @test ret isa IncrementalInference.MsgRelativeType
end

@testset "a solve through the shim" begin
  fg = generateGraph_LineStep(6; poseEvery = 1, landmarkEvery = 0, posePriorsAt = [0])
  getSolverParams(fg).useMsgLikelihoods = true
  tree = solveTree!(fg)
  @test isapprox(getPPE(fg, :x5).suggested[1], 5.0; atol = 1.0)
end

@testset "a sampler table longer than N sends its clique down the generic path" begin
  fg = initfg()
  getSolverParams(fg).N = 50
  addVariable!(fg, :x0, ContinuousScalar)
  addFactor!(fg, [:x0], Prior(AliasingScalarSampler(collect(range(0, 1; length = 120)), ones(120))))
  f = getFactor(fg, :x0f1)
  @test IIFNbpExt.supported(f) && !IIFNbpExt.supported(f, 50) && IIFNbpExt.supported(f, 200)
end
