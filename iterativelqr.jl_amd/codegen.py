"""Model code generation: user function on symbols -> derivatives -> HIP device code.

This is the MI355X-side twin of the reference's Symbolics pipeline
(src/dynamics.jl:16-34, src/costs.jl:17-44, src/constraints.jl:17-43): the user
function is traced on symbolic vectors, Jacobians / gradients / Hessians are
derived symbolically, and instead of `eval(build_function(...))` producing a
Julia closure, a C++ struct of `__device__` functions is emitted that the
solve kernels (csrc/ilqr_device.hpp) are instantiated with.
"""
import hashlib

import sympy as sp
from sympy.printing.c import C99CodePrinter
from sympy.printing.precedence import PRECEDENCE


def _variables(prefix, n):
    return [sp.Symbol("%s%d" % (prefix, i), real=True) for i in range(n)]


class _Traced:
    def _trace(self, f, num_state, num_action, num_parameter):
        self.num_state, self.num_action, self.num_parameter = num_state, num_action, num_parameter
        self.x = _variables("x", num_state)
        self.u = _variables("u", num_action)
        self.w = _variables("w", num_parameter)
        out = f(self.x, self.u, self.w) if num_parameter > 0 else f(self.x, self.u)   # src/dynamics.jl:23
        return out


def _as_list(y):
    if isinstance(y, sp.MatrixBase):
        return [sp.sympify(v) for v in y]
    if isinstance(y, (list, tuple)):
        return [sp.sympify(v) for v in y]
    try:
        import numpy as np
        if isinstance(y, np.ndarray):
            return [sp.sympify(v) for v in y.ravel().tolist()]
    except ImportError:
        pass
    return [sp.sympify(y)]


def _jac(exprs, vars_):
    return [[sp.diff(e, v) for v in vars_] for e in exprs]   # [row][col]


class Dynamics(_Traced):
    """Dynamics(f, num_state, num_action; num_parameter=0) — src/dynamics.jl:16-34."""

    def __init__(self, f, num_state, num_action, num_parameter=0):
        self.evaluate = _as_list(self._trace(f, num_state, num_action, num_parameter))
        self.num_next_state = len(self.evaluate)
        self.jacobian_state = _jac(self.evaluate, self.x)
        self.jacobian_action = _jac(self.evaluate, self.u)


class Cost(_Traced):
    """Cost(f, num_state, num_action; num_parameter=0) — src/costs.jl:17-44."""

    def __init__(self, f, num_state, num_action, num_parameter=0):
        ev = _as_list(self._trace(f, num_state, num_action, num_parameter))
        assert len(ev) == 1, "cost must be scalar"
        self.evaluate = ev[0]
        self.gradient_state = [sp.diff(self.evaluate, v) for v in self.x]
        self.gradient_action = [sp.diff(self.evaluate, v) for v in self.u]
        self.hessian_state_state = _jac(self.gradient_state, self.x)
        self.hessian_action_action = _jac(self.gradient_action, self.u)
        self.hessian_action_state = _jac(self.gradient_action, self.x)


class Constraint(_Traced):
    """Constraint(f, num_state, num_action; indices_inequality, num_parameter) — src/constraints.jl:17-43.

    `Constraint()` (no arguments) is the empty constraint of src/constraints.jl:45-52.
    indices_inequality is 1-based like the reference.
    """

    def __init__(self, f=None, num_state=0, num_action=0, indices_inequality=(), num_parameter=0):
        if f is None:
            self.num_state = self.num_action = self.num_parameter = 0
            self.x, self.u, self.w = [], [], []
            self.evaluate, self.jacobian_state, self.jacobian_action = [], [], []
            self.num_constraint = 0
            self.indices_inequality = []
            return
        self.evaluate = _as_list(self._trace(f, num_state, num_action, num_parameter))
        self.num_constraint = len(self.evaluate)
        self.jacobian_state = _jac(self.evaluate, self.x)
        self.jacobian_action = _jac(self.evaluate, self.u)
        self.indices_inequality = [int(i) for i in indices_inequality]
        assert all(1 <= i <= self.num_constraint for i in self.indices_inequality)


# --------------------------------------------------------------------------- printing
# While the cooperative rollout code of a small model is being printed, its floating-point literals are collected here and
# printed as members of the WaveCtx (cx.k[i], pinned in VGPRs by wave_ctx): hipcc otherwise keeps two dozen fp64 constants in
# scalar register PAIRS, runs out of them, and rebuilds pairs from halves with s_mov every timestep (8 of 150 issue slots
# of the acrobot step). Constants the ISA encodes inline stay literals.
_CONST_CTX = None
_INLINE_FP = {0.0, 0.5, -0.5, 1.0, -1.0, 2.0, -2.0, 4.0, -4.0}


class _Printer(C99CodePrinter):
    # Sums of products are emitted as EXPLICIT fma chains, and every generated function is compiled with FMA contraction off
    # (`#pragma clang fp contract(off)`, _fn below): which product of a sum the backend fuses under hipcc's default
    # -ffp-contract=fast depends on the code the function is inlined into, so the same model function gave results one ulp apart in
    # the latency, the packed and the stage kernels (car_obs: trajectories 5e-14 apart after a hand-over). With the fusion written
    # out, a generated function computes the same bits wherever it is inlined. Order: the terms of a sum in sympy's canonical
    # order; plain terms are added first, then each product is folded in with one fma (a product of more than two factors keeps
    # its first factor as the fma's multiplier and forms the rest with plain multiplications).
    def _print_Add(self, expr):
        plain, prods = [], []
        for t in expr.as_ordered_terms():
            f = [a for a in sp.Mul.make_args(t)]
            nonnum = [a for a in f if not a.is_Number]
            if len(f) >= 2 and not (len(f) == 2 and f[0] == -1) and len(nonnum) >= 1 and not (len(nonnum) == 1 and len(f) == 2 and f[0] in (1, -1)):
                prods.append(f)
            else:
                plain.append(t)
        if not prods:
            return super()._print_Add(expr)
        acc = None
        if plain:
            acc = super()._print_Add(sp.Add(*plain, evaluate=False)) if len(plain) > 1 else self._print(plain[0])
            if len(plain) > 1:
                acc = "(" + acc + ")"
        for f in prods:
            a = f[0]
            rest = sp.Mul(*f[1:], evaluate=False) if len(f) > 2 else f[1]
            sa = self.parenthesize(a, PRECEDENCE["Mul"])
            sr = self._print(rest) if len(f) == 2 else "(" + self._print_Mul_plain(f[1:]) + ")"
            if acc is None:
                acc = "(%s*%s)" % (sa, self.parenthesize(rest, PRECEDENCE["Mul"]) if len(f) == 2 else sr)
            else:
                acc = "fma(%s, %s, %s)" % (sa, sr, acc)
        return acc

    def _print_Mul_plain(self, factors):
        return "*".join(self.parenthesize(a, PRECEDENCE["Mul"]) for a in factors)

    def _print_Float(self, expr):
        v = float(expr)
        if _CONST_CTX is not None and v not in _INLINE_FP:
            if v not in _CONST_CTX:
                _CONST_CTX.append(v)
            return "cx.k[%d]" % _CONST_CTX.index(v)
        return repr(v)

    def _print_Integer(self, expr):
        return "%d.0" % int(expr)

    def _print_Rational(self, expr):
        return "(%d.0/%d.0)" % (expr.p, expr.q)

    def _print_Pow(self, expr):
        b, e = expr.base, expr.exp
        if e.is_Integer and 2 <= int(e) <= 4:
            s = self.parenthesize(b, PRECEDENCE["Mul"])
            return "(" + "*".join([s] * int(e)) + ")"
        if e == -1:
            return "(1.0/%s)" % self.parenthesize(b, PRECEDENCE["Mul"])
        if e.is_Integer and -4 <= int(e) <= -2:
            s = self.parenthesize(b, PRECEDENCE["Mul"])
            return "(1.0/(" + "*".join([s] * (-int(e))) + "))"
        return super()._print_Pow(expr)


_P = _Printer({"strict": False})


def _extract_trig(exprs):
    """Replace every sin(a)/cos(a) by opaque symbols, innermost first, so that one fused
    sincos per distinct argument can be emitted. Returns (exprs, defs) with
    defs = [(arg_expr, s_sym, c_sym, need_s, need_c)] in dependency order."""
    defs = []
    exprs = list(exprs)
    k = 0
    rnd = 0
    while True:
        atoms = set()
        for e in exprs + [d[0] for d in defs]:
            atoms |= e.atoms(sp.sin, sp.cos)
        inner = [a for a in atoms if not a.args[0].has(sp.sin, sp.cos)]
        if not inner:
            break
        by_arg = {}
        for a in inner:
            by_arg.setdefault(a.args[0], set()).add(type(a))
        rep = {}
        for arg in sorted(by_arg, key=sp.default_sort_key):
            s_sym, c_sym = sp.Symbol("sn%d" % k, real=True), sp.Symbol("cs%d" % k, real=True)
            k += 1
            kinds = by_arg[arg]
            defs.append([arg, s_sym, c_sym, sp.sin in kinds, sp.cos in kinds, rnd])
            rep[sp.sin(arg)] = s_sym
            rep[sp.cos(arg)] = c_sym
        exprs = [e.xreplace(rep) for e in exprs]
        for d in defs:
            d[0] = d[0].xreplace(rep)
        rnd += 1
    return exprs, defs


# lanes that cooperate on one dependency level of trig arguments / reciprocals: a whole wave for large models, a 16-lane
# row for small ones (so that the packed kernel can run their rollout with four instances per wave)
COOP_GROUP = 64
MAX_CONSTRAINT_ROWS = 256     # ILQR_MAX_CONSTRAINT_ROWS of include/ilqr_hip.h


def _group_reciprocals(repl):
    """CSE nodes of the form sym = 1/base that do not depend on each other, in groups of 2..64 (coop blocks only)."""
    defs = dict(repl)
    memo = {}

    def deps(sym):
        if sym not in memo:
            memo[sym] = set()
            for f in defs[sym].free_symbols:
                if f in defs:
                    memo[sym] |= {f} | deps(f)
        return memo[sym]

    recs = [(s_, e.base) for s_, e in repl if isinstance(e, sp.Pow) and e.exp == -1]
    groups, used = [], set()
    for i, (si, bi) in enumerate(recs):
        if si in used:
            continue
        g = [(si, bi)]
        for sj, bj in recs[i + 1:]:
            if sj in used or len(g) >= COOP_GROUP:
                continue
            if all(sj not in deps(sk) and sk not in deps(sj) for sk, _ in g):
                g.append((sj, bj))
        if len(g) >= 2:
            groups.append(g)
            used |= {sk for sk, _ in g}
    return groups


# Per-lane constants of the cooperative rollout code of the model being generated (small models): name -> one value per
# cooperating slot. They become members of the model's WaveCtx, built once per kernel by wave_ctx(lane).
_CTX = []


def _ctx_coef(values, nslots=8):
    vals = tuple(float(v) for v in values) + (0.0,) * (nslots - len(values))
    for i, v in enumerate(_CTX):
        if v == vals:
            return i
    _CTX.append(vals)
    return len(_CTX) - 1


def _affine_args(chunk_exprs, repl):
    """Trig arguments of one cooperative batch as affine forms of values every lane already holds: returns
    (symbols, rows) with rows[q] = ([coefficient per symbol], constant) or None if some argument is not affine.
    Symbols are whatever the CSE'd expressions bottom out in (x*, u*, w*, earlier sn*/cs* results)."""
    defs = dict(repl)

    def expand(e):
        while True:
            fs = [f for f in e.free_symbols if f in defs]
            if not fs:
                return e
            e = e.xreplace({f: defs[f] for f in fs})
    full = [sp.expand(expand(e)) for e in chunk_exprs]
    syms = sorted(set().union(*[f.free_symbols for f in full]), key=str)
    rows = []
    for f in full:
        try:
            poly = sp.Poly(f, *syms) if syms else None
        except sp.PolynomialError:
            return None
        if poly is not None and poly.total_degree() > 1:
            return None
        coefs = [sp.sympify(f.coeff(sy, 1)) for sy in syms]
        const = sp.sympify(f.xreplace({sy: 0 for sy in syms}))
        if any(c.free_symbols for c in coefs) or const.free_symbols:
            return None
        if sp.simplify(f - (sum(c * sy for c, sy in zip(coefs, syms)) + const)) != 0:
            return None
        rows.append(([float(c) for c in coefs], float(const)))
    return syms, rows


def _select(var, sel, qi, expr):
    """var = (sel == qi) ? expr : var. For small models the (cheap) expression is evaluated on every lane first and made
    opaque, so that hipcc emits two v_cndmask instead of an exec-masked branch around one addition."""
    if COOP_GROUP == 16:
        return "{ double sel_ = %s; ILQR_OPAQUE(sel_); %s = (%s == %d) ? sel_ : %s; }" % (expr, var, sel, qi, var)
    return "%s = (%s == %d) ? (%s) : %s;" % (var, sel, qi, expr, var)


def _emit_block(outputs, prefix, coop=False):
    """outputs: list of (lhs_string, expr). Returns C statements: CSE temporaries, one
    fused ilqr::sincos_fast per distinct trig argument, then the outputs, in dependency order.

    coop=True emits the WAVE-COOPERATIVE form used on the serial rollout path: all lanes
    hold the same values, so up to four trig arguments of one dependency level are
    evaluated by a single sincos (lane&3 picks the argument) and handed back to every
    lane with DPP quad broadcasts. Needs `const int ql = lane & 3;` in scope."""
    ops_ = [o[2] if len(o) > 2 else "=" for o in outputs]
    outputs = [(o[0], o[1]) for o in outputs]
    exprs = [sp.sympify(e) for _, e in outputs]
    if not exprs:
        return []
    exprs, trig = _extract_trig(exprs)
    nout = len(exprs)
    syms = sp.numbered_symbols(prefix)
    # sympy's "basic" pre/post-optimisations often INCREASE the flop count of mechanical-system
    # expressions (acrobot dynamics: 139 vs 111 ops); run both and keep the cheaper result
    allx = exprs + [d[0] for d in trig]
    cands = []
    for opt in (None, "basic"):
        r, e = sp.cse(allx, symbols=sp.numbered_symbols(prefix), optimizations=opt)
        cost = sum(sp.count_ops(v) for _, v in r) + sum(sp.count_ops(v) for v in e)
        cands.append((cost, r, e))
    _, repl, red = min(cands, key=lambda c: c[0])
    del syms
    # nodes: (defined symbols, expression, text emitter)
    nodes = []
    recip_groups = _group_reciprocals(repl) if coop else []
    grouped = {s for g in recip_groups for s, _ in g}
    for s, e in repl:
        if s in grouped:
            continue
        nodes.append(({s}, e, "const double %s = %s;" % (s, _P.doprint(e))))
    # wave-cooperative reciprocals: mutually independent 1/x nodes are divided on different lanes by ONE division
    # sequence (11 fp64 instructions) and handed back with v_readlane — same IEEE quotient, fewer issue slots
    for gi, g in enumerate(recip_groups):
        txt = ["double rb%d = %s;" % (gi, _P.doprint(g[0][1]))]
        for qi, (_, base) in enumerate(g[1:], start=1):
            txt.append(_select("rb%d" % gi, "l16" if COOP_GROUP == 16 else "lane", qi, _P.doprint(base)))
        # small models (serial rollout chain of the LDS / packed kernels): v_rcp + two Newton steps instead of the IEEE
        # division sequence (5 instead of 11 issue slots per group, <= 1 ulp)
        txt.append(("const double rr%d = ilqr::recip_fast(rb%d);" if COOP_GROUP == 16 else "const double rr%d = 1.0 / rb%d;") % (gi, gi))
        for qi, (sym, _) in enumerate(g):
            txt.append("const double %s = ilqr::wave_bcast<%d>(rr%d);" % (sym, qi, gi))
        nodes.append(({sym for sym, _ in g}, sp.Tuple(*[b for _, b in g]), "\n".join(txt)))
    trig_red = list(zip(trig, red[nout:]))
    if coop:
        rounds = {}
        for d, e in trig_red:
            rounds.setdefault(d[5], []).append((d, e))
        batch_id = 0
        for r in sorted(rounds) if COOP_GROUP == 16 else []:
            # small models: every angle of a dependency level on a PAIR of lanes of the 16-lane row (even lane: sine, odd lane:
            # cosine; ilqr::sincos_pair), up to eight angles per batch; results handed to the whole row by DPP row broadcasts.
            # Affine arguments (the usual case: integrator stages of angles) are formed as one FMA chain with per-lane constant
            # coefficients from the WaveCtx instead of one select per angle.
            items = rounds[r]
            for chunk in [items[i:i + 8] for i in range(0, len(items), 8)]:
                j = batch_id
                batch_id += 1
                aff = _affine_args([e for _, e in chunk], repl)
                txt = []
                if aff is not None:
                    syms_, rows = aff
                    consts = [row[1] for row in rows]
                    acc = None
                    if any(c != 0.0 for c in consts):
                        acc = "cx.a[%d]" % _ctx_coef(consts)
                    for si, sy in enumerate(syms_):
                        col = [row[0][si] for row in rows]
                        if all(c == 0.0 for c in col):
                            continue
                        k_ = "cx.a[%d]" % _ctx_coef(col)
                        acc = ("%s * %s" % (k_, sy)) if acc is None else ("fma(%s, %s, %s)" % (k_, sy, acc))
                    txt.append("const double ta%d = %s;" % (j, acc if acc is not None else "0.0"))
                else:
                    txt.append("double ta%d = %s;" % (j, _P.doprint(chunk[0][1])))
                    for qi, (_, e) in enumerate(chunk[1:], start=1):
                        txt.append(_select("ta%d" % j, "pq", qi, _P.doprint(e)))
                txt.append("const double tr%d = ilqr::sincos_pair(ta%d, cx.tp);" % (j, j))
                defined, dep = set(), sp.Tuple(*[e for _, e in chunk])
                for qi, (d, _) in enumerate(chunk):
                    _, s_sym, c_sym, need_s, need_c, _ = d
                    if need_s:
                        txt.append("const double %s = ilqr::wave_bcast<%d>(tr%d);" % (s_sym, 2 * qi, j))
                    if need_c:
                        txt.append("const double %s = ilqr::wave_bcast<%d>(tr%d);" % (c_sym, 2 * qi + 1, j))
                    defined |= {s_sym, c_sym}
                    d.append("done")
                nodes.append((defined, dep, "\n".join(txt)))
        for r in sorted(rounds) if COOP_GROUP != 16 else []:
            items = rounds[r]
            # up to 4 arguments: one per lane of every quad, results handed back by DPP quad broadcasts
            # (stay in VGPRs); 5..64 arguments: one per lane of the wave, results by v_readlane (SGPRs).
            chunks = [items] if len(items) <= COOP_GROUP else [items[i:i + COOP_GROUP] for i in range(0, len(items), COOP_GROUP)]
            for chunk in chunks:
                if len(chunk) == 1:
                    continue   # a lone argument: plain (wave-uniform) sincos below
                quad = len(chunk) <= 4
                sel, bc = ("ql", "ilqr::quad_bcast") if quad else ("lane", "ilqr::wave_bcast")
                j = batch_id
                batch_id += 1
                txt = ["double ta%d = %s;" % (j, _P.doprint(chunk[0][1]))]
                for qi, (_, e) in enumerate(chunk[1:], start=1):
                    txt.append(_select("ta%d" % j, sel, qi, _P.doprint(e)))
                txt.append("double ts%d, tc%d; ilqr::sincos_fast(ta%d, ts%d, tc%d);" % (j, j, j, j, j))
                defined, dep = set(), sp.Tuple(*[e for _, e in chunk])
                for qi, (d, _) in enumerate(chunk):
                    _, s_sym, c_sym, need_s, need_c, _ = d
                    if need_s:
                        txt.append("const double %s = %s<%d>(ts%d);" % (s_sym, bc, qi, j))
                    if need_c:
                        txt.append("const double %s = %s<%d>(tc%d);" % (c_sym, bc, qi, j))
                    defined |= {s_sym, c_sym}
                    d.append("done")
                nodes.append((defined, dep, "\n".join(txt)))
    for d, e in trig_red:
        if len(d) > 6:
            continue
        _, s_sym, c_sym, need_s, need_c = d[:5]
        a = _P.doprint(e)
        if need_s and need_c:
            txt = "double %s, %s; ilqr::sincos_fast(%s, %s, %s);" % (s_sym, c_sym, a, s_sym, c_sym)
        elif need_s:
            txt = "const double %s = ilqr::sin_fast(%s);" % (s_sym, a)
        else:
            txt = "const double %s = ilqr::cos_fast(%s);" % (c_sym, a)
        nodes.append(({s_sym, c_sym}, e, txt))
    for (lhs, _), e, op in zip(outputs, red[:nout], ops_):
        nodes.append((set(), e, "%s %s %s;" % (lhs, op, _P.doprint(e))))
    defined_by = {}
    for i, (ds, _, _) in enumerate(nodes):
        for s in ds:
            defined_by[s] = i
    lines, done = [], set()

    def visit(i):
        if i in done:
            return
        done.add(i)
        for s in sorted(nodes[i][1].free_symbols, key=str):
            if s in defined_by and defined_by[s] != i:
                visit(defined_by[s])
        lines.extend(nodes[i][2].split("\n"))

    import sys
    sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
    for i in range(len(nodes)):
        visit(i)
    # CSE temporaries that only fed a trig argument now formed from per-lane coefficients are dead: drop them
    import re
    changed = True
    while changed:
        changed = False
        for i, l in enumerate(lines):
            m_ = re.match(r"^const double (%s\d+) = [^;]*;$" % re.escape(prefix), l)
            if m_ and not any(re.search(r"\b%s\b" % m_.group(1), o) for j, o in enumerate(lines) if j != i):
                del lines[i]
                changed = True
                break
    return lines


def _unpack(names_dims):
    lines = []
    for arr, syms in names_dims:
        for i, s in enumerate(syms):
            lines.append("const double %s = %s[%d];" % (s, arr, i))
    return lines


# Every generated function is compiled with FMA contraction off and carries its fusions explicitly (_Printer._print_Add): the same
# bits wherever it is inlined — every kernel family forms the objective in one arithmetic (ilqr_device.hpp: objective_term), and
# an instance may change kernels in the middle of a solve (hand-over), so the linearisation, the rollout and the cost must not
# depend on the inlining context either.
def _fn(ret, name, args, body):
    out = ["    __device__ __forceinline__ static %s %s(%s) {" % (ret, name, ", ".join(args))]
    out.append("#pragma clang fp contract(off)")
    out += ["        " + l for l in body]
    out.append("    }")
    return out


def _arr(name, n, const=True):
    return "%sdouble (&%s)[%d]" % ("const " if const else "", name, max(n, 1))


def _used(lines, sym):
    import re
    pat = re.compile(r"\b%s\b" % sym)
    return any(pat.search(l) for l in lines)


def _prune_unpack(unpack, body):
    """Drop unused `const double x3 = x[3];` lines (keeps -Wunused quiet)."""
    keep = []
    for l in unpack:
        sym = l.split()[2]
        if _used(body, sym):
            keep.append(l)
    return keep


def _table(name, rows):
    """static constexpr double name[R][C] = {...};"""
    body = ",\n".join("        {" + ", ".join(repr(float(v)) for v in r) + "}" for r in rows)
    return ["    static constexpr double %s[%d][%d] = {" % (name, len(rows), len(rows[0])), body, "    };"]


def _itable(name, vals):
    """static constexpr int name[N] = {...};  (at least one element)"""
    vals = list(vals) or [0]
    return ["    static constexpr int %s[%d] = {%s};" % (name, len(vals), ", ".join(str(int(v)) for v in vals))]


def _emit_large_model_extras(L, add, dynamics, n, m, nw, sig_xu):
    """Extra members for models on the HBM-resident large path (ilqr_device_large.hpp):

    * JAC_CONST_FX / JAC_CONST_FU + dyn_jac_var_mem: the Jacobian entries that do not depend on (x, u, w) — for
      mostly-linear models nearly all of them — are written by coalesced wave-wide stores from these tables and
      only the JAC_NVAR state-dependent entries are evaluated and stored per timestep;
    * DYN_AFF + dyn_rem_wave: y = DYN_AFF [x; u; 1] + r(x, u, w). The affine part of row i is evaluated by lane i
      (one FMA chain per lane instead of the whole dense model on every lane); the remainder r keeps the
      wave-cooperative trig form of dyn_wave.
    """
    xs, us = dynamics.x, dynamics.u
    z = list(xs) + list(us)
    # --- Jacobian split
    cfx = [[0.0] * (n * n)]
    cfu = [[0.0] * (n * m)]
    var = []
    for j in range(n):
        for i in range(n):
            e = sp.sympify(dynamics.jacobian_state[i][j])
            if e.free_symbols:
                var.append(("fx[%d]" % (j * n + i), e))
            else:
                cfx[0][j * n + i] = float(e)
    for j in range(m):
        for i in range(n):
            e = sp.sympify(dynamics.jacobian_action[i][j])
            if e.free_symbols:
                var.append(("fu[%d]" % (j * n + i), e))
            else:
                cfu[0][j * n + i] = float(e)
    L.append("    static constexpr int JAC_NVAR = %d;   // state-dependent Jacobian entries (of %d)" % (len(var), n * n + n * m))
    L.extend(_table("JAC_CONST_FX", cfx))
    L.extend(_table("JAC_CONST_FU", cfu))
    # compact form: entry q of dyn_jac_var's output belongs at JAC_VAR_IDX[q] of the concatenation [fx (n*n) | fu (n*m)]
    L.extend(_itable("JAC_VAR_IDX", [int(l[3:-1]) + (n * n if l.startswith("fu") else 0) for l, _ in var]))
    add("void", "dyn_jac_var", sig_xu + [_arr("v", len(var), False)], dynamics, [("v[%d]" % q, e) for q, (_, e) in enumerate(var)])
    # elementwise state-dependent entries: v_q = h(x_{i_q}; w) + c_q with ONE expression h for every q and a constant c_q per entry
    # (e.g. x⁺ = A x + B u + c sin x: the variable entries are A_ii + c cos x_i on the diagonal of fx). The linearisation then
    # evaluates them one (timestep, entry) pair per thread, coalesced, instead of all of a timestep's entries on one thread
    # (JAC_VAR_SRC[q] = i_q, JAC_VAR_ADD[q] = c_q)
    own_j = sp.Symbol("xl", real=True)
    h0, src, addc, jac_elem = None, [], [], len(var) > 0
    for _, e in var:
        e = sp.expand(e)
        fs = [k_ for k_, s_ in enumerate(xs) if s_ in e.free_symbols]
        if len(fs) != 1 or any(s_ in e.free_symbols for s_ in us):
            jac_elem = False
            break
        cq, rest = e.as_independent(*e.free_symbols, as_Add=True)
        hq = rest.subs(xs[fs[0]], own_j)
        if h0 is None:
            h0 = hq
        elif hq != h0:
            jac_elem = False
            break
        src.append(fs[0])
        addc.append(float(cq))
    L.append("    static constexpr bool JAC_VAR_ELEMENTWISE = %s;" % ("true" if jac_elem else "false"))
    if jac_elem:
        L.extend(_itable("JAC_VAR_SRC", src))
        L.extend(_table("JAC_VAR_ADD", [addc]))
        add("double", "dyn_jac_var_own", ["const double xl", _arr("w", nw)], dynamics, [], ret_expr=h0)
    # --- affine split of the dynamics
    aff = [[0.0] * (n + m + 1) for _ in range(n)]
    rem = []
    zi = {s_: k for k, s_ in enumerate(z)}
    for i, y in enumerate(dynamics.evaluate):
        r = 0
        for term in sp.Add.make_args(sp.expand(sp.sympify(y))):
            if not term.free_symbols:
                aff[i][n + m] += float(term)
                continue
            c, rest = term.as_coeff_Mul()
            if rest in zi and c.is_number:
                aff[i][zi[rest]] += float(c)
            else:
                r = r + term
        rem.append(r)
    L.extend(_table("DYN_AFF", aff))
    L.append("    static constexpr bool DYN_HAS_REM = %s;" % ("true" if any(sp.sympify(r) != 0 for r in rem) else "false"))
    L.append("#if defined(__HIPCC__)")
    add("void", "dyn_rem_wave", ["const int lane"] + sig_xu + [_arr("r", n, False)], dynamics,
        [("r[%d]" % i, e) for i, e in enumerate(rem)], coop=True)
    L.append("#endif")
    # elementwise remainder: r_i = g(x_i; u, w) with ONE expression g for every row (e.g. x⁺ = A x + B u + c sin x): the lane that
    # owns row i evaluates g on its own state component, no cross-lane traffic at all
    own = sp.Symbol("xl", real=True)
    g0, elementwise = None, any(sp.sympify(r) != 0 for r in rem)
    for i, r in enumerate(rem):
        r = sp.sympify(r)
        if any(s_ in r.free_symbols for k_, s_ in enumerate(xs) if k_ != i):
            elementwise = False
            break
        gi = r.subs(xs[i], own)
        if g0 is None:
            g0 = gi
        elif gi != g0:
            elementwise = False
            break
    L.append("    static constexpr bool DYN_REM_ELEMENTWISE = %s;" % ("true" if elementwise else "false"))
    add("double", "dyn_rem_own", ["const double xl", _arr("u", m), _arr("w", nw)], dynamics, [], ret_expr=(g0 if elementwise else sp.Integer(0)))



def _emit_compact_hessians(L, add, add_al_c, al_terms, cost_stage, cost_term, con_stage, con_term, dynamics, n, m, ncs, nct, sig_xu, sig_x):
    """Large path: the accumulated cost Hessians (reference quirk Q1, src/costs.jl:74) live in HBM as ONE compact row per
    timestep holding only the structurally non-zero entries  [gxx | guu | gux]  (union of the cost Hessians and of the
    Gauss-Newton AL terms of src/gradients.jl:54-80; terminal cost / constraint entries are part of the gxx set). The gxx
    entries are sorted by the 16x16 MFMA tile they fall in (HESS_XX_TILE_START), so that the wave that owns a tile of Qxx adds
    exactly its own entries. HESS_IDX[q] is the column-major index inside the entry's matrix."""
    import re
    TN = (n + 15) // 16
    outs_s = [("gxx[%d]" % (j * n + i), cost_stage.hessian_state_state[i][j]) for j in range(n) for i in range(n)]
    outs_s += [("guu[%d]" % (j * m + i), cost_stage.hessian_action_action[i][j]) for j in range(m) for i in range(m)]
    outs_s += [("gux[%d]" % (j * m + i), cost_stage.hessian_action_state[i][j]) for j in range(n) for i in range(m)]
    outs_s = [(l, e, "+=") for l, e in outs_s if sp.sympify(e) != 0]
    outs_t = [("gxx[%d]" % (j * n + i), cost_term.hessian_state_state[i][j]) for j in range(n) for i in range(n)]
    outs_t = [(l, e, "+=") for l, e in outs_t if sp.sympify(e) != 0]
    al_s_o, al_s_unp = al_terms(con_stage, ncs, True) if ncs else ([], [])
    al_t_o, al_t_unp = al_terms(con_term, nct, False) if nct else ([], [])
    sets = {"gxx": set(), "guu": set(), "gux": set()}
    for l, _, _ in outs_s + outs_t + al_s_o + al_t_o:
        k_, idx = l[:3], l[4:-1]
        if k_ in sets:
            sets[k_].add(int(idx))

    def tile_of(idx):
        col, row = divmod(idx, n)
        return (row // 16) * TN + (col // 16)

    xx = sorted(sets["gxx"], key=lambda e: (tile_of(e), e))
    uu, ux = sorted(sets["guu"]), sorted(sets["gux"])
    starts = [0] * (TN * TN + 1)
    for e in xx:
        starts[tile_of(e) + 1] += 1
    for q in range(TN * TN):
        starts[q + 1] += starts[q]
    pos = {("gxx", e): q for q, e in enumerate(xx)}
    pos.update({("guu", e): len(xx) + q for q, e in enumerate(uu)})
    pos.update({("gux", e): len(xx) + len(uu) + q for q, e in enumerate(ux)})
    L.append("    // compact structural Hessian row: [gxx (tile-sorted) | guu | gux]")
    L.append("    static constexpr int HESS_NXX = %d, HESS_NUU = %d, HESS_NUX = %d;" % (len(xx), len(uu), len(ux)))
    L.extend(_itable("HESS_IDX", xx + uu + ux))
    L.extend(_itable("HESS_XX_TILE_START", starts))

    def compact(outs_):
        res = []
        for o in outs_:
            l = o[0]
            k_ = l[:3]
            if k_ in sets:
                res.append(("hs[%d]" % pos[(k_, int(l[4:-1]))],) + tuple(o[1:]))
            else:
                res.append(o)
        return res

    add("void", "cost_s_hess_c", sig_xu + ["double* __restrict__ hs"], cost_stage, compact(outs_s))
    add("void", "cost_t_hess_c", sig_x + ["double* __restrict__ hs"], cost_term, compact(outs_t), with_u=False)
    add_al_c("al_s_c", sig_xu + [_arr("ct", ncs), _arr("ir", ncs), _arr("gx", n, False), _arr("gu", m, False), "double* __restrict__ hs"],
             con_stage if ncs else dynamics, compact(al_s_o), al_s_unp, True)
    add_al_c("al_t_c", sig_x + [_arr("ct", nct), _arr("ir", nct), _arr("gx", n, False), "double* __restrict__ hs"],
             con_term if nct else dynamics, compact(al_t_o), al_t_unp, False)


def generate_model_source(name, dynamics, cost_stage, cost_term, con_stage=None, con_term=None):
    """Return (struct_name, C++ source) of the device model struct.

    dynamics: Dynamics; cost_stage/cost_term: Cost (terminal has num_action == 0);
    con_stage/con_term: Constraint or None (= empty constraint).
    Matrices are emitted column-major to match Julia / the C-ABI.
    """
    con_stage = con_stage if con_stage is not None else Constraint()
    con_term = con_term if con_term is not None else Constraint()
    n, m, nw = dynamics.num_state, dynamics.num_action, dynamics.num_parameter
    assert dynamics.num_next_state == n, "time-uniform state dimension required"
    assert cost_stage.num_state == n and cost_stage.num_action == m
    assert cost_term.num_state == n and cost_term.num_action == 0
    ncs, nct = con_stage.num_constraint, con_term.num_constraint
    assert ncs <= MAX_CONSTRAINT_ROWS and nct <= MAX_CONSTRAINT_ROWS, "at most %d constraint rows per stage" % MAX_CONSTRAINT_ROWS
    assert n <= 64 and m <= 16, "device kernels: nx <= 64 (one state component per lane on the large path), nu <= 16"
    ineq_s = sum(1 << (i - 1) for i in con_stage.indices_inequality)
    ineq_t = sum(1 << (i - 1) for i in con_term.indices_inequality)
    sname = "Model_" + name
    global COOP_GROUP
    COOP_GROUP = 16 if (n <= 4 and m <= 4) else 64
    L = []
    L.append("// GENERATED by iterativelqr.jl_amd/codegen.py — do not edit.")
    L.append("// Device model functions (value + symbolic derivatives), column-major outputs.")
    L.append("struct %s {" % sname)
    L.append("    static constexpr int NX = %d, NU = %d, NW = %d, NCS = %d, NCT = %d;" % (n, m, nw, ncs, nct))
    L.append("    static constexpr unsigned long long INEQ_S = 0x%xull, INEQ_T = 0x%xull;" % (ineq_s & (2 ** 64 - 1), ineq_t & (2 ** 64 - 1)))
    if ncs > 64 or nct > 64:
        # more rows than a 64-bit mask holds: the masks as words (ilqr::IneqMask picks them up; models with <= 64 rows keep their header)
        nwords = (max(ncs, nct) + 63) // 64
        words = lambda v: ", ".join("0x%xull" % ((v >> (64 * k)) & (2 ** 64 - 1)) for k in range(nwords))
        L.append("    static constexpr int INEQ_WORDS = %d;" % nwords)
        L.append("    static constexpr unsigned long long INEQ_S_W[%d] = {%s}, INEQ_T_W[%d] = {%s};" % (nwords, words(ineq_s), nwords, words(ineq_t)))
    L.append('    static constexpr const char* NAME = "%s";' % name)

    def rename(obj):
        # every traced object has its own x*/u*/w* symbols with identical names, so no renaming needed
        return obj

    xs, us, ws = dynamics.x, dynamics.u, dynamics.w
    sig_xu = [_arr("x", n), _arr("u", m), _arr("w", nw)]
    sig_x = [_arr("x", n), _arr("w", nw)]

    def unpack_xu(obj, with_u=True):
        items = [("x", [str(s) for s in obj.x])]
        if with_u:
            items.append(("u", [str(s) for s in obj.u]))
        items.append(("w", [str(s) for s in obj.w]))
        return _unpack(items)

    def add(ret, fname, sig, obj, outputs, with_u=True, ret_expr=None, coop=False):
        body = _emit_block(outputs + ([("const double ret_", ret_expr)] if ret_expr is not None else []), "t", coop=coop)
        if ret_expr is not None:
            body.append("return ret_;")
        un = _prune_unpack(unpack_xu(obj, with_u), body)
        if coop:
            un = [("const int l16 = lane & 15, pq = l16 >> 1; (void)l16; (void)pq;" if COOP_GROUP == 16
                   else "const int ql = lane & 3; (void)ql;")] + un
        L.extend(_fn(ret, fname, sig, un + body))

    # dynamics
    add("void", "dyn", sig_xu + [_arr("y", n, False)], dynamics,
        [("y[%d]" % i, e) for i, e in enumerate(dynamics.evaluate)])
    # wave-cooperative variant for the serial closed-loop rollout (all lanes hold the same x, u)
    L.append("#if defined(__HIPCC__)")
    del _CTX[:]
    mark = len(L)
    global _CONST_CTX
    _CONST_CTX = [] if COOP_GROUP == 16 else None
    add("void", "dyn_wave", (["const WaveCtx& cx"] if COOP_GROUP == 16 else []) + ["const int lane"] + sig_xu + [_arr("y", n, False)], dynamics,
        [("y[%d]" % i, e) for i, e in enumerate(dynamics.evaluate)], coop=True)
    consts, _CONST_CTX = _CONST_CTX, None
    if COOP_GROUP == 16:
        # per-lane constants of the cooperative code, built ONCE per kernel (not per timestep): the trig pair coefficients and
        # the coefficient columns of the affine trig arguments, indexed by the angle slot pq = (lane & 15) >> 1
        ctx = ["    struct WaveCtx { ilqr::TrigPair tp; double a[%d], k[%d]; };" % (max(1, len(_CTX)), max(1, len(consts))),
               "    // PIN_CONSTANTS = true makes the model's fp64 constants opaque scalar-register pairs: hipcc otherwise keeps halves of known",
               "    // 64-bit constants and re-assembles aligned pairs with s_mov_b32 at every use (8 of 148.5 issue slots of an acrobot rollout step)",
               "    template <bool PIN_CONSTANTS = false> __device__ __forceinline__ static WaveCtx wave_ctx(const int lane) {",
               "        const int pq = (lane & 15) >> 1; (void)pq;",
               "        WaveCtx cx;",
               "        cx.tp = ilqr::make_trig_pair<PIN_CONSTANTS>(lane);"]
        if not _CTX:
            ctx.append("        cx.a[0] = 0.0;")
        for i, vals in enumerate(_CTX):
            expr = repr(vals[-1])
            for q in range(len(vals) - 2, -1, -1):
                expr = "(pq == %d) ? %r : (%s)" % (q, vals[q], expr)
            ctx.append("        cx.a[%d] = %s; ILQR_OPAQUE(cx.a[%d]);" % (i, expr, i))
        if not consts:
            ctx.append("        cx.k[0] = 0.0;")
        for i, v in enumerate(consts):
            ctx.append("        cx.k[%d] = %r; if (PIN_CONSTANTS) ILQR_OPAQUE_UNIFORM(cx.k[%d]);" % (i, v, i))
        ctx += ["        return cx;", "    }"]
        L[mark:mark] = ctx
    L.append("#endif")
    outs = [("fx[%d]" % (j * n + i), dynamics.jacobian_state[i][j]) for j in range(n) for i in range(n)]
    outs += [("fu[%d]" % (j * n + i), dynamics.jacobian_action[i][j]) for j in range(m) for i in range(n)]
    add("void", "dyn_jac", sig_xu + [_arr("fx", n * n, False), _arr("fu", n * m, False)], dynamics, outs)
    # stage cost
    add("double", "cost_s", sig_xu, cost_stage, [], ret_expr=cost_stage.evaluate)
    outs = [("gx[%d]" % i, e) for i, e in enumerate(cost_stage.gradient_state)]
    outs += [("gu[%d]" % i, e) for i, e in enumerate(cost_stage.gradient_action)]
    add("void", "cost_s_grad", sig_xu + [_arr("gx", n, False), _arr("gu", m, False)], cost_stage, outs)
    outs = [("gxx[%d]" % (j * n + i), cost_stage.hessian_state_state[i][j]) for j in range(n) for i in range(n)]
    outs += [("guu[%d]" % (j * m + i), cost_stage.hessian_action_action[i][j]) for j in range(m) for i in range(m)]
    outs += [("gux[%d]" % (j * m + i), cost_stage.hessian_action_state[i][j]) for j in range(n) for i in range(m)]
    add("void", "cost_s_hess", sig_xu + [_arr("gxx", n * n, False), _arr("guu", m * m, False), _arr("gux", m * n, False)],
        cost_stage, outs)
    # terminal cost
    add("double", "cost_t", sig_x, cost_term, [], with_u=False, ret_expr=cost_term.evaluate)
    add("void", "cost_t_grad", sig_x + [_arr("gx", n, False)], cost_term,
        [("gx[%d]" % i, e) for i, e in enumerate(cost_term.gradient_state)], with_u=False)
    add("void", "cost_t_hess", sig_x + [_arr("gxx", n * n, False)], cost_term,
        [("gxx[%d]" % (j * n + i), cost_term.hessian_state_state[i][j]) for j in range(n) for i in range(n)], with_u=False)
    # constraints
    if ncs > 0:
        assert con_stage.num_state == n and con_stage.num_action == m
    add("void", "con_s", sig_xu + [_arr("c", ncs, False)], con_stage if ncs else dynamics,
        [("c[%d]" % i, e) for i, e in enumerate(con_stage.evaluate)])
    outs = [("cx[%d]" % (j * ncs + i), con_stage.jacobian_state[i][j]) for j in range(n) for i in range(ncs)]
    outs += [("cu[%d]" % (j * ncs + i), con_stage.jacobian_action[i][j]) for j in range(m) for i in range(ncs)]
    add("void", "con_s_jac", sig_xu + [_arr("cx", ncs * n, False), _arr("cu", ncs * m, False)],
        con_stage if ncs else dynamics, outs)
    if nct > 0:
        assert con_term.num_state == n
    add("void", "con_t", sig_x + [_arr("c", nct, False)], con_term if nct else dynamics,
        [("c[%d]" % i, e) for i, e in enumerate(con_term.evaluate)], with_u=False)
    add("void", "con_t_jac", sig_x + [_arr("cx", nct * n, False)], con_term if nct else dynamics,
        [("cx[%d]" % (j * nct + i), con_term.jacobian_state[i][j]) for j in range(n) for i in range(nct)], with_u=False)
    # ---- memory-streaming variants for large models (HBM-resident workspace): Jacobians are
    # written straight to memory, Hessians are ACCUMULATED (`.+=`, src/costs.jl:74) skipping
    # structural zeros, and the Gauss-Newton AL terms of src/gradients.jl:54-80 are derived
    # symbolically so that sparse constraint Jacobians cost nothing.
    def nz(pairs):
        return [(l, e, "+=") for l, e in pairs if sp.sympify(e) != 0]

    large = n > 4 or m > 4
    if not large:
        outs = [("fx[%d]" % (j * n + i), dynamics.jacobian_state[i][j]) for j in range(n) for i in range(n)]
        outs += [("fu[%d]" % (j * n + i), dynamics.jacobian_action[i][j]) for j in range(m) for i in range(n)]
        add("void", "dyn_jac_mem", sig_xu + ["double* __restrict__ fx", "double* __restrict__ fu"], dynamics, outs)
        outs = nz([("gxx[%d]" % (j * n + i), cost_stage.hessian_state_state[i][j]) for j in range(n) for i in range(n)])
        outs += nz([("guu[%d]" % (j * m + i), cost_stage.hessian_action_action[i][j]) for j in range(m) for i in range(m)])
        outs += nz([("gux[%d]" % (j * m + i), cost_stage.hessian_action_state[i][j]) for j in range(n) for i in range(m)])
        add("void", "cost_s_hess_acc", sig_xu + ["double* __restrict__ gxx", "double* __restrict__ guu", "double* __restrict__ gux"], cost_stage, outs)
        outs = nz([("gxx[%d]" % (j * n + i), cost_term.hessian_state_state[i][j]) for j in range(n) for i in range(n)])
        add("void", "cost_t_hess_acc", sig_x + ["double* __restrict__ gxx"], cost_term, outs, with_u=False)

    def al_terms(con, nc, stage):
        ct = [sp.Symbol("ct%d" % i, real=True) for i in range(nc)]     # λ + Iρ c
        ir = [sp.Symbol("ir%d" % i, real=True) for i in range(nc)]     # ρ ∘ a
        cx, cu = con.jacobian_state, con.jacobian_action
        o = []
        for j in range(n):
            o.append(("gx[%d]" % j, sum(cx[i][j] * ct[i] for i in range(nc))))
        for j in range(n):
            for i2 in range(n):
                o.append(("gxx[%d]" % (j * n + i2), sum(cx[i][i2] * ir[i] * cx[i][j] for i in range(nc))))
        if stage:
            for j in range(m):
                o.append(("gu[%d]" % j, sum(cu[i][j] * ct[i] for i in range(nc))))
            for j in range(m):
                for i2 in range(m):
                    o.append(("guu[%d]" % (j * m + i2), sum(cu[i][i2] * ir[i] * cu[i][j] for i in range(nc))))
            for j in range(n):
                for i2 in range(m):
                    o.append(("gux[%d]" % (j * m + i2), sum(cu[i][i2] * ir[i] * cx[i][j] for i in range(nc))))
        unp = ["const double ct%d = ct[%d];" % (i, i) for i in range(nc)] + ["const double ir%d = ir[%d];" % (i, i) for i in range(nc)]
        return nz(o), unp

    def add_al(fname, sig, obj, con, nc, stage, with_u):
        if nc == 0:
            L.extend(_fn("void", fname, sig, []))
            return
        outs_, unp = al_terms(con, nc, stage)
        body = _emit_block(outs_, "t")
        un = _prune_unpack(unpack_xu(obj, with_u) + unp, body)
        L.extend(_fn("void", fname, sig, un + body))

    def add_al_c(fname, sig, obj, outs_, unp, with_u):
        """AL Gauss-Newton terms with the Hessian part in compact structural form (outs_ already renamed)."""
        if not outs_:
            L.extend(_fn("void", fname, sig, []))
            return
        body = _emit_block(outs_, "t")
        un = _prune_unpack(unpack_xu(obj, with_u) + unp, body)
        L.extend(_fn("void", fname, sig, un + body))

    if large:
        _emit_large_model_extras(L, add, dynamics, n, m, nw, sig_xu)
        _emit_compact_hessians(L, add, add_al_c, al_terms, cost_stage, cost_term, con_stage, con_term, dynamics, n, m, ncs, nct, sig_xu, sig_x)
    else:
        add_al("al_s", sig_xu + [_arr("ct", ncs), _arr("ir", ncs), _arr("gx", n, False), _arr("gu", m, False),
                                 "double* __restrict__ gxx", "double* __restrict__ guu", "double* __restrict__ gux"], con_stage if ncs else dynamics, con_stage, ncs, True, True)
        add_al("al_t", sig_x + [_arr("ct", nct), _arr("ir", nct), _arr("gx", n, False), "double* __restrict__ gxx"],
               con_term if nct else dynamics, con_term, nct, False, False)
    L.append("};")
    src = "\n".join(L) + "\n"
    # the cooperative rollout code is generic over HOW a value travels from one lane of the cooperating group to all of
    # them: ilqr::RowBC (16-lane rows, one v_mov_b64_dpp row_newbcast per value) for small models in every kernel — four
    # instances per wave in the packed kernel, four identical copies of one instance in the LDS kernels — and ilqr::WaveBC
    # (whole wave, v_readlane) for the remainder code of large models
    src = src.replace("ilqr::wave_bcast<", "BC::template bcast<")
    for fn in ("dyn_wave", "dyn_rem_wave"):
        src = src.replace("__device__ __forceinline__ static void %s(" % fn,
                          "template <class BC = ilqr::%s> __device__ __forceinline__ static void %s(" % ("RowBC" if COOP_GROUP == 16 else "WaveBC", fn))
    return sname, src


def source_hash(src):
    return hashlib.sha256(src.encode()).hexdigest()[:16]


# --------------------------------------------------------------------------- symbolic objects as C callables (the C-source seam)
def c_callable(name, exprs, obj, terminal=False):
    """One `ILQR_MODEL_FN void name(double* out, const double* x, const double* u, const double* w)` of the reference's in-place
    callable contract (src/dynamics.jl:55-60, src/costs.jl:1-15, src/constraints.jl:54-64) from symbolic expressions — what the Julia
    wrapper gets from Symbolics.build_function(...; target = CTarget()). `exprs`: flat list in the callable's COLUMN-MAJOR output
    order; zero entries are not written (out arrives zeroed). Plain C (math.h), no dependence on the device headers."""
    from sympy.printing.c import C99CodePrinter
    pr = C99CodePrinter({"strict": False})
    exprs = [sp.sympify(e) for e in exprs]
    repl, red = sp.cse(exprs, symbols=sp.numbered_symbols("t")) if exprs else ([], [])
    sub = {s_: sp.Symbol("x[%d]" % i) for i, s_ in enumerate(obj.x)}
    if not terminal:
        sub.update({s_: sp.Symbol("u[%d]" % i) for i, s_ in enumerate(obj.u)})
    sub.update({s_: sp.Symbol("w[%d]" % i) for i, s_ in enumerate(obj.w)})
    body = ["    const double %s = %s;" % (sym, pr.doprint(e.xreplace(sub))) for sym, e in repl]
    body += ["    out[%d] = %s;" % (i, pr.doprint(e.xreplace(sub))) for i, e in enumerate(red) if e != 0]
    return "ILQR_MODEL_FN void %s(double* out, const double* x, const double* u, const double* w) {\n%s\n}\n" % (name, "\n".join(body))


def _colmajor(rows):
    """[row][col] -> flat column-major list"""
    return [rows[i][j] for j in range(len(rows[0]) if rows else 0) for i in range(len(rows))]


def c_dynamics(prefix, d):
    return (c_callable(prefix, d.evaluate, d) + c_callable(prefix + "_jacobian_state", _colmajor(d.jacobian_state), d) +
            c_callable(prefix + "_jacobian_action", _colmajor(d.jacobian_action), d))


def c_cost(prefix, c, terminal=False):
    src = (c_callable(prefix, [c.evaluate], c, terminal) + c_callable(prefix + "_gradient_state", c.gradient_state, c, terminal) +
           c_callable(prefix + "_hessian_state_state", _colmajor(c.hessian_state_state), c, terminal))
    if not terminal:
        src += (c_callable(prefix + "_gradient_action", c.gradient_action, c) +
                c_callable(prefix + "_hessian_action_action", _colmajor(c.hessian_action_action), c) +
                c_callable(prefix + "_hessian_action_state", _colmajor(c.hessian_action_state), c))
    return src


def c_constraint(prefix, k, terminal=False):
    src = c_callable(prefix, k.evaluate, k, terminal) + c_callable(prefix + "_jacobian_state", _colmajor(k.jacobian_state), k, terminal)
    if not terminal:
        src += c_callable(prefix + "_jacobian_action", _colmajor(k.jacobian_action), k)
    return src
