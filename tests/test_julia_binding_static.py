"""Static check of the Julia binding (julia/AriannaHIP.jl) against the C header (include/amc.h).

There is no Julia in this image or on the GPU box, so the ccall stub cannot be executed.  What CAN be checked without
Julia is everything a ccall gets wrong silently: the field list of the `AmcConfig` mirror (names, order, widths) against
`struct amc_config`, and for every `ccall((:amc_..., libamc), ...)` the entry point's existence, its return type, its
argument types (C type -> the Julia types that are ABI-compatible with it) and the number of values actually passed.
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "julia", "src", "AriannaHIP.jl")
JL_RUN = os.path.join(ROOT, "julia", "src", "AriannaHIPRun.jl")          # include()d by the module: run_fused!, deferred stores
JL_TEST = os.path.join(ROOT, "julia", "test", "runtests.jl")             # the package's tests: their ccalls are checked like the module's
HDR = os.path.join(ROOT, "include", "amc.h")

# C type (normalised: no `const`, single spaces, `*` attached) -> Julia types with the same calling-convention class
C_TO_JULIA = {
    "amc_handle*": {"Ptr{Cvoid}"},
    "amc_handle**": {"Ref{Ptr{Cvoid}}", "Ptr{Ptr{Cvoid}}"},
    "amc_config*": {"Ref{AmcConfig}", "Ptr{AmcConfig}"},
    "char*": {"Cstring", "Ptr{UInt8}", "Ptr{Cchar}"},
    "char**": {"Ptr{Cstring}", "Ptr{Ptr{UInt8}}"},
    "double*": {"Ptr{Float64}", "Ref{Float64}"},
    "double": {"Float64", "Cdouble"},
    "float": {"Float32", "Cfloat"},
    "int": {"Cint", "Int32"},
    "int*": {"Ptr{Cint}", "Ref{Cint}", "Ptr{Int32}"},
    "int64_t": {"Int64"},
    "int64_t*": {"Ptr{Int64}", "Ref{Int64}"},
    "uint64_t": {"UInt64"},
    "uint64_t*": {"Ptr{UInt64}", "Ref{UInt64}"},
    "uint32_t": {"UInt32"},
    "uint32_t*": {"Ptr{UInt32}"},
    "void*": {"Ptr{Cvoid}"},
    "void**": {"Ref{Ptr{Cvoid}}", "Ptr{Ptr{Cvoid}}"},
}
C_FIELD_TO_JULIA = {"uint32_t": "UInt32", "int32_t": "Int32", "int64_t": "Int64", "uint64_t": "UInt64", "double": "Float64",
                    "double*": "Ptr{Float64}", "void*": "Ptr{Cvoid}"}
C_RET_TO_JULIA = {"int": {"Cint", "Int32"}, "char*": {"Cstring", "Ptr{UInt8}"}}


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def _norm_ctype(t):
    t = re.sub(r"\bconst\b", " ", t)
    t = re.sub(r"\s*\*\s*", "*", t.strip())
    return re.sub(r"\s+", " ", t).strip()


def header_prototypes():
    text = _strip_c_comments(open(HDR).read())
    protos = {}
    for m in re.finditer(r"^\s*(const\s+char\s*\*\s*|int\s+)(amc_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.M | re.S):
        ret, name, args = _norm_ctype(m.group(1)), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)(\w+)$", a, flags=re.S)            # last identifier is the parameter name
                params.append(_norm_ctype(mm.group(1)))
        protos[name] = (ret, params)
    return protos


def header_config_fields():
    text = _strip_c_comments(open(HDR).read())
    body = re.search(r"typedef\s+struct\s+amc_config\s*\{(.*?)\}\s*amc_config\s*;", text, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        mm = re.match(r"(.*?)(\w+)$", decl, flags=re.S)
        fields.append((mm.group(2), _norm_ctype(mm.group(1))))
    return fields


def _split_top(s):
    """Split on commas that are not inside (), {} or []."""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def julia_ccalls():
    # Julia line comments (no '#' inside the strings used there); docstrings name no ccall
    text = "\n".join(re.sub(r"#[^\n]*", "", open(f).read()) for f in (JL, JL_RUN, JL_TEST))
    calls = []
    for m in re.finditer(r"ccall\(", text):
        i = m.end()
        depth, j = 1, i
        while depth:
            depth += {"(": 1, ")": -1}.get(text[j], 0)
            j += 1
        parts = _split_top(text[i:j - 1])
        target = re.match(r"\(\s*:(\w+)\s*,\s*libamc\s*\)", parts[0])
        assert target, f"ccall target not of the form (:amc_x, libamc): {parts[0]}"
        argt = parts[2].strip()
        assert argt.startswith("(") and argt.endswith(")"), argt
        types = _split_top(argt[1:-1])
        calls.append(dict(name=target.group(1), ret=parts[1].strip(), types=[t for t in types if t], values=parts[3:]))
    return calls


def julia_config_fields():
    text = open(JL).read()
    body = re.search(r"^struct AmcConfig\n(.*?)^end", text, flags=re.S | re.M).group(1)
    fields = []
    for line in body.splitlines():
        line = line.split("#")[0].strip()
        if line:
            name, typ = line.split("::")
            fields.append((name.strip(), typ.strip()))
    return fields


def test_amc_config_mirror_matches_the_header_field_for_field():
    c_fields, j_fields = header_config_fields(), julia_config_fields()
    assert [n for n, _ in j_fields] == [n for n, _ in c_fields], "AmcConfig: field names / order differ from struct amc_config"
    for (name, ctype), (_, jtype) in zip(c_fields, j_fields):
        assert C_FIELD_TO_JULIA[ctype] == jtype, f"AmcConfig.{name}: {jtype} does not mirror C `{ctype}`"
    # the ABI guard the library checks (struct_size) is what sizeof gives for this layout: 16 fields, natural alignment
    size = {"UInt32": 4, "Int32": 4, "Int64": 8, "UInt64": 8, "Float64": 8, "Ptr{Float64}": 8, "Ptr{Cvoid}": 8}
    off = 0
    for _, t in j_fields:
        off = (off + size[t] - 1) // size[t] * size[t] + size[t]
    from montecarlo_amd import _capi
    import ctypes
    assert (off + 7) // 8 * 8 == ctypes.sizeof(_capi.AmcConfig)


def test_every_ccall_matches_its_prototype():
    protos = header_prototypes()
    assert len(protos) >= 40 and "amc_sweep" in protos and "amc_last_error" in protos
    calls = julia_ccalls()
    assert len(calls) >= 15
    for c in calls:
        assert c["name"] in protos, f"ccall to :{c['name']}, which include/amc.h does not declare"
        ret, params = protos[c["name"]]
        assert c["ret"] in C_RET_TO_JULIA[ret], f":{c['name']} returns C `{ret}`, the ccall says {c['ret']}"
        assert len(c["types"]) == len(params), f":{c['name']} takes {len(params)} arguments, the ccall declares {len(c['types'])}"
        assert len(c["values"]) == len(params), f":{c['name']}: {len(c['values'])} values passed for {len(params)} parameters"
        for k, (ctype, jtype) in enumerate(zip(params, c["types"])):
            assert jtype in C_TO_JULIA[ctype], f":{c['name']} argument {k + 1}: C `{ctype}` bound as {jtype}"


def test_binding_covers_the_entry_points_a_run_needs():
    used = {c["name"] for c in julia_ccalls()}
    need = {"amc_create", "amc_create_custom", "amc_create_model", "amc_create_policy_model", "amc_destroy",
            "amc_upload_state", "amc_sweep", "amc_download_state", "amc_download_counters", "amc_reduce_begin",
            # sums cross the shards as exact records, merged and rounded once (reproducible sums): never as rounded doubles
            "amc_reduce_end_exact", "amc_allreduce_xsum", "amc_xsum_round", "amc_pg_estimate_exact",
            "amc_create_proposal_model", "amc_create_vector_policy_model", "amc_n_params",
            "amc_comm_unique_id", "amc_comm_init", "amc_set_parameters",
            "amc_get_parameters", "amc_pgmc_steps", "amc_pg_get_accumulated", "amc_pg_set_accumulated", "amc_last_error"}
    assert need <= used, sorted(need - used)


def test_shards_are_connected_and_pools_are_not_deep_copied_per_chain():
    """The two defects of the first stub: no amc_comm_init anywhere (amc_allreduce_sum was a no-op on N > 1) and one
    deepcopy of the pool per chain (1e7 copies at the headline size; metropolis.jl:289 does that, but the engine keeps
    the per-chain part on the device)."""
    text = re.sub(r"#[^\n]*", "", open(JL).read())
    assert "comm_init!(alg, rank, n_ranks, unique_id)" in text
    assert not re.search(r"\[\s*deepcopy\(pool\)\s+for\s+\w+\s+in\s+chains\s*\]", text)
    assert text.count("deepcopy(pool)") == 1                    # the one template copy
    # one parameters object per move, shared by the update and every materialised pools[c]
    assert "move.policy, move.parameters, move.weight" in text and "est.parameters_list" in text


def test_philox_rng_stub_follows_the_current_arithmetic_spec():
    """julia/PhiloxRNG.jl restates the draw schedule for the reference's R= hook; it cannot run here, so at least its
    bit-field constants must be the ones the oracle implements (spec v5: 28-bit angle, 12 + 12 spare bits per chain,
    36-bit pick uniform, 12 + 40-bit accept uniform) -- a spec change that forgets the Julia file fails here."""
    text = re.sub(r"#[^\n]*", "", open(os.path.join(ROOT, "julia", "src", "PhiloxRNG.jl")).read())
    src = open(os.path.join(ROOT, "oracle", "amc_oracle.c")).read()
    for jl, c in [("Float64(w >> 4) * 2.0^-27", "(double)(w >> 4) * 0x1.0p-27"),
                  ("(v[3] >> 12) & 0x00000fff", "((v[2] >> 12) & 0xFFFu)"),
                  ("(v[3] >> 24) | ((v[4] & 0x0000000f) << 8)", "((v[2] >> 24) | ((v[3] & 0xFu) << 8))"),
                  ("(UInt64(pick12 & 0x00000fff) << 24) | UInt64(lo & 0x00ffffff)) * 2.0^-36", "0x1.0p-36"),
                  ("(UInt64(accept12 & 0x00000fff) << 40) | (((UInt64(hi) << 32) | lo) >> 24)", "<< 40) | ((((uint64_t)hi << 32) | lo) >> 24)")]:
        assert jl in text, jl
        assert c in src, c
    assert "angle_oc2" not in text and "sincospi_tab(angle28(v[4]))" in text


def test_julia_host_has_the_fused_run_loop():
    """`north_star` names Julia as the host: the look-ahead over the schedulers, the callback sums formed in the observed launch,
    whole [Metropolis, estimator, update] stretches in one engine call and the deferred rows exist on the Julia side too
    (julia/AriannaHIPRun.jl restates montecarlo_amd/simulation.py run(fuse=True)), and the module includes them."""
    main = open(JL).read()
    run = re.sub(r"#[^\n]*", "", open(JL_RUN).read())
    assert 'include("AriannaHIPRun.jl")' in main
    for name in ("function run_fused!(simulation::Simulation)", "function fuse_sweeps!", "function fuse_pgmc!", "function make_step_observed!",
                 "mutable struct HIPStoreCallbacks", "mutable struct HIPStoreParameters", "struct HIPDeviceEstimator", "struct HIPDeviceUpdate",
                 "function declare_reduction_needs!", "function consecutive(", "function observed_next("):
        assert name in run, name
    used = {c["name"] for c in julia_ccalls()}
    need = {"amc_sweep_reduce_begin", "amc_pgmc_steps_reduce_begin", "amc_pgmc_steps", "amc_parameters_begin", "amc_parameters_end_all",
            "amc_set_reduce_columns", "amc_pg_accumulate", "amc_pg_update", "amc_sync", "amc_reduce_end_exact"}
    assert need <= used, sorted(need - used)
    # a stretch nobody observes is ONE amc_sweep of n sweeps, not n calls of one
    assert re.search(r"ccall\(\(:amc_sweep, libamc\), Cint, \(Ptr\{Cvoid\}, Int64\), alg\.handle, n\)", run)
    # the loop keeps the reference's order: initialise all, steps, finalise all in a `finally` (src/simulation.jl:175-204)
    body = run[run.index("function run_fused!"):]
    assert body.index("initialise(algorithm, simulation)") < body.index("while t <= simulation.steps") < body.index("finally") < body.index("finalise(algorithm, simulation)")
    # every `function` / `struct` / control block of the file is closed (a cheap stand-in for a parser)
    opens = len(re.findall(r"^\s*(?:mutable struct|struct|function|if|for|while|try|begin|let)\b|\bdo\b\s*(?:\([^)]*\)|\w+)?\s*$|=\s*begin\s*$|@elapsed begin", run, flags=re.M))
    closes = len(re.findall(r"^\s*end\b", run, flags=re.M))
    assert opens == closes, (opens, closes)


def _balanced(text):
    """Every `function` / `struct` / control block of a Julia file is closed (a cheap stand-in for a parser)."""
    text = re.sub(r'"""(?:.|\n)*?"""', '""', text)               # docstrings
    text = re.sub(r"#[^\n]*", "", text)
    text = re.sub(r'"(?:[^"\\\n]|\\.)*"', '""', text)            # string literals (no interpolated blocks with keywords in these files)
    opens = 0
    for line in text.splitlines():
        if re.match(r"\s*(?:[\w.]*@[\w.]+\s+(?:\w+\s+)*?)?(?:mutable struct|struct|function|if|for|while|try|begin|let|module|macro|quote)\b", line):
            opens += 1                 # a block keyword opens the line (possibly behind a macro: `@inbounds for`, `GC.@preserve a b begin`)
        elif re.search(r"\bdo\b\s*(?:\([^)]*\)|[\w, ]+)?\s*$|\bbegin\s*$|=\s*let\b[^\n]*$", line):
            opens += 1                 # ... or a block opens at its end (`map(xs) do x`, `@testset "..." begin`, `x = let y = ...`)
    closes = len(re.findall(r"^\s*end\b", text, flags=re.M))
    return opens, closes


def test_julia_is_a_package_with_its_tests():
    """SURVEY section 8 row f3 asks for PhiloxRNG for the reference's R= hook AND the package's tests: julia/ is a package
    (Project.toml, src/, test/runtests.jl) a maintainer runs with one command; the tests replay the committed
    golden trajectories through STOCK Metropolis(...; R=PhiloxRNG{seed,1}) -- the only route from "statistically pinned" to
    pinned --, through HIPMetropolis where a GPU exists, and check test/ad_backends_test.jl's closed forms.  No Julia here: what
    can be checked is that the pieces exist, name what they must, and hang together."""
    proj = open(os.path.join(ROOT, "julia", "Project.toml")).read()
    assert re.search(r'^name = "AriannaHIP"$', proj, flags=re.M) and re.search(r'^uuid = "[0-9a-f-]{36}"$', proj, flags=re.M)
    deps = proj[proj.index("[deps]"):proj.index("[compat]")]
    assert 'Arianna = "07692032-97b4-4f8d-80d7-e18df88d31a9"' in deps and "Random = " in deps and "Libdl = " in deps
    ref_toml = "/root/reference/Project.toml"
    if os.path.exists(ref_toml):           # the uuid IS the reference package's
        assert 'uuid = "07692032-97b4-4f8d-80d7-e18df88d31a9"' in open(ref_toml).read()
    extras = proj[proj.index("[extras]"):]
    for pkg in ("ComponentArrays", "Distributions", "JSON", "Test", "Statistics"):
        assert re.search(rf"^{pkg} = ", extras, flags=re.M), pkg
        assert f'"{pkg}"' in extras[extras.index("[targets]"):], pkg
    main = open(JL).read()
    assert "module AriannaHIP" in main and 'include("PhiloxRNG.jl")' in main and "using .PhiloxRNGs: PhiloxRNG" in main
    assert re.search(r"^export .*HIPMetropolis.*PhiloxRNG", main, flags=re.M) and "function available()" in main
    for f in ("AriannaHIP.jl", "AriannaHIPRun.jl", "PhiloxRNG.jl", "amc_tables.jl"):
        assert os.path.exists(os.path.join(ROOT, "julia", "src", f)), f
    t = open(JL_TEST).read()
    # (i) stock Metropolis through the reference's R= hook on the reference's own model file, replaying cases 0-3
    assert 'include(joinpath(pkgdir(Arianna), "example", "particle_1d", "particle_1d.jl"))' in t
    assert "algorithm=Metropolis" in t and "R=PhiloxRNG{seed,1}" in t and 'GOLDEN["cases"][1:4]' in t
    assert "oracle_trajectories.json" in t and "reference_kats.json" in t
    # ... and the estimator's hook (estimator.jl:63,92) on the state the sweeps left, the reference's aliased scratch buffer accounted for
    assert "PolicyGradientEstimator(chains; dependencies=(metropolis,)" in t and "R=PhiloxRNG{seed,2}" in t
    assert 'case["x_after_pg"]' in t and 'case["pg_estimate_q3"]' in t and "want[3, k] - d[1] + d[2]" in t
    # (ii) the device path through the binding, skipped -- and said so -- without libamc / a GPU
    assert "algorithm=AriannaHIP.HIPMetropolis" in t and "AriannaHIP.available()" in t and "@test_skip" in t
    # (iii) ad_backends closed forms, through amc_selftest_math where a device exists
    assert ":amc_selftest_math" in t and 'KATS["ad_backends"]' in t and "withgrad_log_proposal_density!" in t
    # mismatches are reported, not hidden: the flip note is printed AND the test fails
    assert "@info" in t and "@test isempty(off)" in t
    for path in (JL_TEST, JL, JL_RUN, os.path.join(ROOT, "julia", "src", "PhiloxRNG.jl")):
        opens, closes = _balanced(open(path).read())
        assert opens == closes, (path, opens, closes)
    # the fixtures the Julia test reads hold what it indexes
    import json
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_trajectories.json")))
    names = [c["spec"]["name"] for c in golden["cases"][:4]]
    assert names == ["harmonic_K1", "harmonic_K2_pgmc_pool", "double_well_K2", "harmonic_K1_shard_offset"]
    for c in golden["cases"][:4]:
        assert c["snapshots"][0]["sweep"] == 0 and {"x", "e", "accepted", "total", "energy", "acceptance"} <= set(c["snapshots"][1])
        assert "dtype" not in c["spec"] and "proposal" not in c["spec"] and "classes" not in c["spec"]      # plain particle_1d
        assert len(c["pg_estimate_q3"]) == 5 * len(c["spec"]["sigma"]) and len(c["x_after_pg"]) == c["spec"]["M"]
    kats = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")))
    assert {"delta", "sigma", "logq", "grad_sigma", "atol"} <= set(kats["ad_backends"])


def test_hexfloat_of_the_julia_tests_reads_every_fixture_value():
    """runtests.jl parses the fixtures' C99 hex floats with a regular expression of its own: every value of the four cases it
    replays must match it, and the exact reconstruction mantissa x 2^(exponent - 4 digits) must give the value back."""
    import json
    rx = re.search(r'm = match\(r"(\^[^"]+\$)", s\)', open(JL_TEST).read()).group(1)
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_trajectories.json")))
    n = 0
    for c in golden["cases"][:4]:
        for snap in c["snapshots"]:
            for key in ("x", "e", "acceptance"):
                for text in snap.get(key, []):
                    if text in ("nan", "inf", "-inf"):
                        continue
                    m = re.match(rx, text)
                    assert m, text
                    frac = m.group(3)
                    mant = int(m.group(2) + frac, 16)
                    assert mant < 2 ** 53
                    val = float(mant) * 2.0 ** (int(m.group(4)) - 4 * len(frac))
                    assert (-val if m.group(1) else val) == float.fromhex(text)
                    n += 1
    assert n > 500
