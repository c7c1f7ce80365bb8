# HipKKTExt.jl -- Clarabel.jl-side binding of libhipkkt.so (include/hipkkt.h).
#
# NOT executed in the build container (Julia is not installed there: SURVEY.md F4); the call sequences of B and C are
# executed by their tested Python mirrors (cuclarabel_amd/ipm.py: HipBackend, HipSystemBackend(lazy=True, host_cones=True)).  Written
# against Clarabel.jl v0.11.0 following the pattern of its own out-of-tree backends
# (ext/directldl_hsl.jl, ext/directldl_pardiso.jl).  Three bindings:
#
#   A.  HipDirectLDLSolver <: AbstractDirectLDLSolver     select with
#           Clarabel.Settings(direct_solve_method = :hipldl)
#       smallest glue, solver.jl / kktsystem.jl / kktsolver_directldl.jl untouched.
#   B.  HipKKTSolver <: AbstractKKTSolver                  moves assembly, regularisation and
#       iterative refinement to the device as well; needs the one-line dispatch in
#       src/kktsystem.jl:33 shown at the bottom (that call site hard-codes DirectLDLKKTSolver).
#   C.  HipKKTSystem <: AbstractKKTSystem                  the reduced-system layer on the device too (right-hand-side
#       construction, the three solves per iteration, step recovery); the constant and the affine right-hand side share
#       one 2-column solve although solver.jl still issues kkt_update! and kkt_solve!(:affine) as two calls (lazy
#       mode); needs the one-line dispatch in src/solver.jl:139 shown at the bottom.
#
# Load as a package extension ([weakdeps]/[extensions] in Project.toml, as HSLExt is) or simply
# `include` it after `using Clarabel`.  ENV["HIPKKT_LIB"] may point at libhipkkt.so.

module HipKKTExt

using SparseArrays, Clarabel
import Clarabel: DefaultInt, AbstractDirectLDLSolver, AbstractKKTSolver, LinearSolverInfo, Settings
import Clarabel: ldlsolver_constructor, ldlsolver_matrix_shape, ldlsolver_is_available
import Clarabel: linear_solver_info, update_values!, scale_values!, refactor!, solve!
import Clarabel: kktsolver_update!, kktsolver_setrhs!, kktsolver_solve!, kktsolver_update_P!,
                 kktsolver_update_A!, kktsolver_linear_solver_info
import Clarabel: CompositeCone, ZeroCone, NonnegativeCone, SecondOrderCone, PSDTriangleCone,
                 get_Hs!, is_sparse_expandable, numel

const libhipkkt = get(ENV, "HIPKKT_LIB", "libhipkkt.so")

# mirrors hipkkt_settings (include/hipkkt.h); field order and types must match
struct CSettings
    static_regularization_constant::Cdouble
    static_regularization_proportional::Cdouble
    dynamic_regularization_eps::Cdouble
    dynamic_regularization_delta::Cdouble
    iterative_refinement_reltol::Cdouble
    iterative_refinement_abstol::Cdouble
    iterative_refinement_stop_ratio::Cdouble
    iterative_refinement_max_iter::Int32
    static_regularization_enable::Int32
    iterative_refinement_enable::Int32
    ordering::Int32
    nd_leaf_size::Int32
    device::Int32
    user_perm::Ptr{Int64}
    amd_dense_scale::Cdouble
end

csettings(s::Settings{Float64}) = CSettings(
    s.static_regularization_constant, s.static_regularization_proportional,
    s.dynamic_regularization_eps, s.dynamic_regularization_delta,
    s.iterative_refinement_reltol, s.iterative_refinement_abstol, s.iterative_refinement_stop_ratio,
    s.iterative_refinement_max_iter, s.static_regularization_enable, s.iterative_refinement_enable,
    1 #= HIPKKT_ORDER_ND =#, 1000, -1, C_NULL, 1.5)

struct CInfo
    n::Int64; m::Int64; p::Int64; N::Int64; nnzK::Int64; nnzL::Int64; nnzL_stored::Int64
    nsuper::Int64; nlevels::Int64; max_front::Int64; etree_height::Int64; nHs::Int64
    nsparse_soc::Int64; sparse_soc_len::Int64
    factor_flops::Cdouble; front_bytes::Cdouble; update_bytes::Cdouble
end

last_error() = unsafe_string(ccall((:hipkkt_last_error, libhipkkt), Cstring, ()))

# 0 -> true, > 0 (numeric failure) -> false, < 0 -> error   (the reference's Bool convention)
function check(rc::Cint, what)
    rc == 0 && return true
    rc > 0 && return false
    error("$what failed ($rc): $(last_error())")
end

# ------------------------------------------------------------------ A. direct LDL backend
mutable struct HipDirectLDLSolver{T} <: AbstractDirectLDLSolver{T}
    handle::Ptr{Cvoid}
    function HipDirectLDLSolver{T}(KKT::SparseMatrixCSC{T}, Dsigns, settings) where {T}
        T === Float64 || error("hipldl supports Float64 only")
        h = Ref{Ptr{Cvoid}}(C_NULL)
        cs = Ref(csettings(settings))
        ds = Vector{Int64}(Dsigns)
        rc = GC.@preserve KKT ds ccall((:hipkkt_ldl_create, libhipkkt), Cint,
            (Ref{Ptr{Cvoid}}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Cdouble}, Ptr{Int64}, Ref{CSettings}, Cint),
            h, KKT.n, KKT.colptr, KKT.rowval, KKT.nzval, ds, cs, 1 #= Julia is 1-based =#)
        check(rc, "hipkkt_ldl_create") || error("hipkkt_ldl_create: numeric failure")
        obj = new(h[])
        finalizer(o -> ccall((:hipkkt_ldl_destroy, libhipkkt), Cvoid, (Ptr{Cvoid},), o.handle), obj)
        return obj
    end
end

ldlsolver_constructor(::Val{:hipldl}) = HipDirectLDLSolver
ldlsolver_matrix_shape(::Val{:hipldl}) = :triu
function ldlsolver_is_available(::Val{:hipldl})     # must not throw (directldl_defaults.jl:22-27)
    try
        return ccall((:hipkkt_available, libhipkkt), Cint, ()) == 1
    catch
        return false
    end
end

function linear_solver_info(s::HipDirectLDLSolver{T}) where {T}
    info = Ref{CInfo}()
    check(ccall((:hipkkt_ldl_info, libhipkkt), Cint, (Ptr{Cvoid}, Ref{CInfo}), s.handle, info), "hipkkt_ldl_info")
    LinearSolverInfo(:hipldl, 1, true, info[].nnzK, info[].nnzL)
end

function update_values!(s::HipDirectLDLSolver{T}, index::AbstractVector{DefaultInt}, values::Vector{T}) where {T}
    idx = index isa Vector{Int64} ? index : collect(Int64, index)      # views of the data maps are common
    GC.@preserve idx values check(ccall((:hipkkt_ldl_update_values, libhipkkt), Cint,
        (Ptr{Cvoid}, Ptr{Int64}, Ptr{Cdouble}, Int64), s.handle, idx, values, length(idx)), "hipkkt_ldl_update_values")
end

function scale_values!(s::HipDirectLDLSolver{T}, index::AbstractVector{DefaultInt}, scale::T) where {T}
    idx = index isa Vector{Int64} ? index : collect(Int64, index)
    GC.@preserve idx check(ccall((:hipkkt_ldl_scale_values, libhipkkt), Cint,
        (Ptr{Cvoid}, Ptr{Int64}, Cdouble, Int64), s.handle, idx, scale, length(idx)), "hipkkt_ldl_scale_values")
end

# K is not needed: the library holds the values pushed through update_values!/scale_values!
refactor!(s::HipDirectLDLSolver{T}, K::SparseMatrixCSC{T}) where {T} =
    check(ccall((:hipkkt_ldl_refactor, libhipkkt), Cint, (Ptr{Cvoid},), s.handle), "hipkkt_ldl_refactor")

function solve!(s::HipDirectLDLSolver{T}, K::SparseMatrixCSC{T}, x::Vector{T}, b::Vector{T}) where {T}
    GC.@preserve x b check(ccall((:hipkkt_ldl_solve, libhipkkt), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), s.handle, x, b), "hipkkt_ldl_solve")
end

# ------------------------------------------------------------------ B. whole KKT solver
mutable struct HipKKTSolver{T} <: AbstractKKTSolver{T}
    handle::Ptr{Cvoid}
    m::DefaultInt
    n::DefaultInt
    Hsblocks::Vector{T}          # get_Hs! target, as in DirectLDLKKTSolver
    soc_u::Vector{T}
    soc_v::Vector{T}
    soc_eta2::Vector{T}

    function HipKKTSolver{T}(P::SparseMatrixCSC{T}, A::SparseMatrixCSC{T}, cones::CompositeCone{T},
                             m::DefaultInt, n::DefaultInt, settings::Settings{T}) where {T}
        T === Float64 || error("hipkkt supports Float64 only")
        kinds = Int32[]; dims = Int64[]
        for c in cones
            if     c isa ZeroCone         push!(kinds, 0); push!(dims, numel(c))
            elseif c isa NonnegativeCone  push!(kinds, 1); push!(dims, numel(c))
            elseif c isa SecondOrderCone  push!(kinds, 2); push!(dims, numel(c))
            elseif c isa PSDTriangleCone  push!(kinds, 3); push!(dims, c.n)
            else error("hipkkt: cone type $(typeof(c)) is not supported")
            end
        end
        h = Ref{Ptr{Cvoid}}(C_NULL)
        cs = Ref(csettings(settings))
        Pt = triu(P)                                # data.P is already triu; harmless otherwise
        rc = GC.@preserve Pt A kinds dims ccall((:hipkkt_kkt_create, libhipkkt), Cint,
            (Ref{Ptr{Cvoid}}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Cdouble},
             Ptr{Int64}, Ptr{Int64}, Ptr{Cdouble}, Int64, Ptr{Int32}, Ptr{Int64}, Ref{CSettings}, Cint),
            h, n, m, Pt.colptr, Pt.rowval, Pt.nzval, A.colptr, A.rowval, A.nzval,
            length(kinds), kinds, dims, cs, 1)
        check(rc, "hipkkt_kkt_create") || error("hipkkt_kkt_create: numeric failure")
        info = Ref{CInfo}()
        check(ccall((:hipkkt_kkt_info, libhipkkt), Cint, (Ptr{Cvoid}, Ref{CInfo}), h[], info), "hipkkt_kkt_info")
        obj = new(h[], m, n, zeros(T, info[].nHs), zeros(T, info[].sparse_soc_len),
                  zeros(T, info[].sparse_soc_len), zeros(T, info[].nsparse_soc))
        finalizer(o -> ccall((:hipkkt_kkt_destroy, libhipkkt), Cvoid, (Ptr{Cvoid},), o.handle), obj)
        return obj
    end
end

function kktsolver_update!(ks::HipKKTSolver{T}, cones::CompositeCone{T}) where {T}
    # same data the reference reads: get_Hs! (kktsolver_directldl.jl:223) and, per sparse SOC,
    # u, v, eta^2 (directldl_datamaps.jl:61-79)
    get_Hs!(cones, ks.Hsblocks)
    off = 0; k = 0
    for c in cones
        if c isa SecondOrderCone && is_sparse_expandable(c)
            d = numel(c)
            ks.soc_u[off+1:off+d] .= c.sparse_data.u
            ks.soc_v[off+1:off+d] .= c.sparse_data.v
            k += 1; ks.soc_eta2[k] = c.η^2
            off += d
        end
    end
    GC.@preserve ks check(ccall((:hipkkt_kkt_update_cones, libhipkkt), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
        ks.handle, ks.Hsblocks, ks.soc_u, ks.soc_v, ks.soc_eta2), "hipkkt_kkt_update_cones")
end

function kktsolver_setrhs!(ks::HipKKTSolver{T}, rhsx::AbstractVector{T}, rhsz::AbstractVector{T}) where {T}
    x = rhsx isa Vector{T} ? rhsx : collect(rhsx); z = rhsz isa Vector{T} ? rhsz : collect(rhsz)
    GC.@preserve x z check(ccall((:hipkkt_kkt_setrhs, libhipkkt), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), ks.handle, x, z), "hipkkt_kkt_setrhs")
    return nothing
end

function kktsolver_solve!(ks::HipKKTSolver{T}, lhsx::Union{Nothing,AbstractVector{T}},
                          lhsz::Union{Nothing,AbstractVector{T}}) where {T}
    px = isnothing(lhsx) ? Ptr{Cdouble}(C_NULL) : pointer(lhsx)
    pz = isnothing(lhsz) ? Ptr{Cdouble}(C_NULL) : pointer(lhsz)
    GC.@preserve lhsx lhsz check(ccall((:hipkkt_kkt_solve, libhipkkt), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), ks.handle, px, pz), "hipkkt_kkt_solve")
end

# Not part of the reference interface: nrhs right-hand sides against the current factorisation in one call
# (columns of RHSX (n x k), RHSZ (m x k)); every column goes through the refinement rule of
# kktsolver_directldl.jl:389-449 on its own.  Returns (is_success, refinement rounds per column).
function kktsolver_solve_multi!(ks::HipKKTSolver{T}, RHSX::Matrix{T}, RHSZ::Matrix{T},
                                LHSX::Union{Nothing,Matrix{T}}, LHSZ::Union{Nothing,Matrix{T}}) where {T}
    k = size(RHSX, 2)
    ir = zeros(Int64, max(k, 1))
    px = LHSX === nothing ? Ptr{Cdouble}(C_NULL) : pointer(LHSX)
    pz = LHSZ === nothing ? Ptr{Cdouble}(C_NULL) : pointer(LHSZ)
    ok = GC.@preserve RHSX RHSZ LHSX LHSZ ir check(ccall((:hipkkt_kkt_solve_multi, libhipkkt), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Int64}),
        ks.handle, k, RHSX, RHSZ, px, pz, ir), "hipkkt_kkt_solve_multi")
    return ok, ir[1:k]
end

kktsolver_update_P!(ks::HipKKTSolver{T}, P::SparseMatrixCSC{T}) where {T} =
    (GC.@preserve P check(ccall((:hipkkt_kkt_update_P, libhipkkt), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), ks.handle, P.nzval), "hipkkt_kkt_update_P"); nothing)
kktsolver_update_A!(ks::HipKKTSolver{T}, A::SparseMatrixCSC{T}) where {T} =
    (GC.@preserve A check(ccall((:hipkkt_kkt_update_A, libhipkkt), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), ks.handle, A.nzval), "hipkkt_kkt_update_A"); nothing)

function kktsolver_linear_solver_info(ks::HipKKTSolver{T}) where {T}
    info = Ref{CInfo}()
    check(ccall((:hipkkt_kkt_info, libhipkkt), Cint, (Ptr{Cvoid}, Ref{CInfo}), ks.handle, info), "hipkkt_kkt_info")
    LinearSolverInfo(:hipkkt, 1, true, info[].nnzK, info[].nnzL)
end

# ------------------------------------------------------------------ C. the reduced-system layer (level C)
# HipKKTSystem <: AbstractKKTSystem replaces DefaultKKTSystem (src/kktsystem.jl:5-232): the three solves per iteration,
# their right-hand-side construction and the recovery of (dx, dz, ds, dtau, dkappa) run on the device.  solver.jl's
# loop is untouched: it keeps calling kkt_update!, kkt_solve!(…, :affine), kkt_solve!(…, :combined) separately
# (solver.jl:278-295, :317-323).  The handle is put in LAZY mode: kkt_update! scales, scatters and refactors and leaves
# (x2, z2) = K \ (-q, b) to the affine kkt_solve!, which sends both right-hand sides through the sweeps as ONE 2-column
# solve and returns the AND of the two statuses -- what `is_kkt_solve_success = kkt_update!(...)` followed by
# `is_kkt_solve_success && kkt_solve!(...)` computes from the two calls.
import Clarabel: AbstractKKTSystem, DefaultProblemData, DefaultVariables
import Clarabel: kkt_update!, kkt_solve_initial_point!, kkt_solve!, kkt_update_P!, kkt_update_A!, kkt_linear_solver_info

mutable struct HipKKTSystem{T} <: AbstractKKTSystem{T}
    kktsolver::HipKKTSolver{T}       # owns the handle (and the get_Hs! / sparse-SOC staging vectors)
    w::Vector{T}                     # NT scaling handed to the device with every kkt_update!: w (m), eta (per cone),
    eta::Vector{T}                   #   lambda (m), R / Rinv of the PSD cones (column-major, concatenated)
    lambda::Vector{T}
    psd_R::Vector{T}
    psd_Rinv::Vector{T}
    tk::Vector{T}                    # (dtau, dkappa) read-back
    variables_sent::Bool             # (x, s, z) of this iteration are on the device already (set by the affine kkt_solve!)
    registered::Vector{Vector{T}}    # host arrays page-locked for the copies (hipkkt_host_register), kept for the finalizer

    function HipKKTSystem{T}(data::DefaultProblemData{T}, cones::CompositeCone{T}, settings::Settings{T}) where {T}
        ks = HipKKTSolver{T}(data.P, data.A, cones, data.m, data.n, settings)
        GC.@preserve data check(ccall((:hipkkt_kkt_system_init, libhipkkt), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}),
            ks.handle, data.q, data.b), "hipkkt_kkt_system_init")
        check(ccall((:hipkkt_kkt_system_set_lazy, libhipkkt), Cint, (Ptr{Cvoid}, Cint), ks.handle, 1), "hipkkt_kkt_system_set_lazy")
        npsd = sum(c isa PSDTriangleCone ? c.n^2 : 0 for c in cones; init = 0)
        obj = new(ks, ones(T, data.m), ones(T, length(cones)), zeros(T, data.m), zeros(T, npsd), zeros(T, npsd), zeros(T, 2),
                  false, Vector{T}[])
        for v in (obj.w, obj.lambda) register!(obj, v) end
        finalizer(o -> foreach(v -> ccall((:hipkkt_host_unregister, libhipkkt), Cint, (Ptr{Cvoid},), v), o.registered), obj)
        return obj
    end
end

# Page-lock the vectors that cross PCIe every iteration.  The solver's DefaultVariables (variables, the step, the two
# right-hand sides) are allocated once per Solver (solver.jl:150-160) and live as long as it does, like this object's
# own scaling vectors: registered the first time they are seen.  Optional -- a vector that cannot be registered is
# copied through the runtime's pageable path as before.
function register!(s::HipKKTSystem{T}, v::Vector{T}) where {T}
    (isempty(v) || any(r -> r === v, s.registered)) && return
    rc = ccall((:hipkkt_host_register, libhipkkt), Cint, (Ptr{Cvoid}, Int64), v, sizeof(v))
    rc == 0 && push!(s.registered, v)
    return
end
function register_once!(s::HipKKTSystem{T}, vs::DefaultVariables{T}...) where {T}
    length(s.registered) >= 2 + 3 * length(vs) && return
    for d in vs, v in (d.x, d.s, d.z) register!(s, v) end
end

kkt_linear_solver_info(s::HipKKTSystem{T}) where {T} = kktsolver_linear_solver_info(s.kktsolver)
kkt_update_P!(s::HipKKTSystem{T}, P::SparseMatrixCSC{T}) where {T} = kktsolver_update_P!(s.kktsolver, P)
kkt_update_A!(s::HipKKTSystem{T}, A::SparseMatrixCSC{T}) where {T} = kktsolver_update_A!(s.kktsolver, A)

# kkt_update! gets the CONES, not the iterate (solver.jl:279): what the device needs is the NT scaling -- fields of the
# cone objects (cone_types.jl:40-60, 84-115, 125-160): w, eta, lambda, and R / Rinv of the PSD cones.  get_Hs!'s blocks
# and the sparse second-order cones' u / v / eta^2 are functions of those (coneops_nncone.jl:91-101,
# coneops_socone.jl:125-192, coneops_psdtrianglecone.jl:135-161) and are formed on the device
# (hipkkt_kkt_system_update_scaling): for the 100k-variable SOCP 3.2 MB cross PCIe per iteration instead of 6.4, and the
# host skips get_Hs! altogether.  (hipkkt_kkt_system_update_cones, which takes get_Hs!'s output as well, remains for a
# caller that wants the reference's own rounding of those blocks.)
function kkt_update!(s::HipKKTSystem{T}, data::DefaultProblemData{T}, cones::CompositeCone{T}) where {T}
    ks = s.kktsolver
    zoff = 0; poff = 0
    for (i, c) in enumerate(cones)
        d = numel(c)
        if c isa NonnegativeCone
            s.w[zoff+1:zoff+d] .= c.w;  s.lambda[zoff+1:zoff+d] .= c.λ
        elseif c isa SecondOrderCone
            s.w[zoff+1:zoff+d] .= c.w;  s.lambda[zoff+1:zoff+d] .= c.λ;  s.eta[i] = c.η
        elseif c isa PSDTriangleCone
            n = c.n
            s.lambda[zoff+1:zoff+n] .= c.data.λ
            s.psd_R[poff+1:poff+n*n]    .= vec(c.data.R)
            s.psd_Rinv[poff+1:poff+n*n] .= vec(c.data.Rinv)
            poff += n * n
        end
        zoff += d
    end
    s.variables_sent = false         # a new iteration: the next kkt_solve! sends (x, s, z) again
    GC.@preserve s ks check(ccall((:hipkkt_kkt_system_update_scaling, libhipkkt), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
        ks.handle, s.w, s.eta, s.lambda, s.psd_R, s.psd_Rinv), "hipkkt_kkt_system_update_scaling")
end

function kkt_solve_initial_point!(s::HipKKTSystem{T}, variables::DefaultVariables{T}, data::DefaultProblemData{T}) where {T}
    GC.@preserve variables check(ccall((:hipkkt_kkt_system_solve_initial_point_host, libhipkkt), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
        s.kktsolver.handle, variables.x, variables.s, variables.z), "hipkkt_kkt_system_solve_initial_point_host")
end

function kkt_solve!(s::HipKKTSystem{T}, lhs::DefaultVariables{T}, rhs::DefaultVariables{T}, data::DefaultProblemData{T},
                    variables::DefaultVariables{T}, cones::CompositeCone{T}, steptype::Symbol) where {T}
    # `variables` does not change between the affine and the combined kkt_solve! of an iteration (solver.jl:289-323):
    # it goes over PCIe with the first of them only (NULL = "the variables of the previous call").
    register_once!(s, lhs, rhs, variables)
    vx, vs, vz = s.variables_sent ? (C_NULL, C_NULL, C_NULL) : (pointer(variables.x), pointer(variables.s), pointer(variables.z))
    ok = GC.@preserve s lhs rhs variables check(ccall((:hipkkt_kkt_system_solve_host, libhipkkt), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
         Cdouble, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Cint),
        s.kktsolver.handle, lhs.x, lhs.s, lhs.z, s.tk, rhs.x, rhs.s, rhs.z, rhs.τ, rhs.κ,
        vx, vs, vz, variables.τ, variables.κ, steptype === :affine ? 0 : 1),
        "hipkkt_kkt_system_solve_host")
    s.variables_sent = true
    ok || return false
    lhs.τ = s.tk[1]
    lhs.κ = s.tk[2]
    return true
end

end # module

# ---- boundary B only: the one-line dispatch at src/kktsystem.jl:33 ("Always LDL for now")
#
#   kktsolver = settings.direct_solve_method === :hipkkt ?
#       HipKKTExt.HipKKTSolver{T}(data.P, data.A, cones, m, n, settings) :
#       DirectLDLKKTSolver{T}(data.P, data.A, cones, m, n, settings)
#
# ---- boundary C: the one-line dispatch at src/solver.jl:139, inside setup! (outside the IPM loop, like boundary B's)
#
#   s.kktsystem = s.settings.direct_solve_method === :hipkkt ?
#       HipKKTExt.HipKKTSystem{T}(s.data, s.cones, s.settings) :
#       DefaultKKTSystem{T}(s.data, s.cones, s.settings)
#
# solver.jl:278-323 (kkt_update!, kkt_solve! :affine, kkt_solve! :combined), :389-394 (default start) and
# data_updating.jl:67,97 then reach the methods above by dispatch on the type of s.kktsystem.
