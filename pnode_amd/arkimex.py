"""IMEX (additive Runge-Kutta) stepping of the reference's semi-implicit path:
``setupTS(u, funcIM, ..., method="imex", implicit_form=True, imex_form=True, func2=funcEX)``
(reference ``pnode/petsc_adjoint.py`` ("pa.py") 600-614 parameter split [IM, EX], 655-656 TS
type ARKIMEX, 414-441 ``evalIFunction`` F = udot - funcIM, 393-412 ``evalRHSFunction`` funcEX,
279-334 ``IJacPShell``; linear_solver="torch": ``pnode/torch_linearsolve.py:7-35``) and its
discrete adjoint.

PETSc's TSARKIMEX restated on the C ABI (identity mass matrix):
    Z_i  = u_n + h sum_{j<i} (At_ij KI_j + A_ij KE_j)
    Y_i  = Z_i                     if At_ii == 0
           solves shift (Y - Z_i) - fI(t_i, Y) = 0,  shift = 1/(h At_ii)      otherwise
    KI_i = shift (Y_i - Z_i)  (= fI(t_i, Y_i) at convergence; fI itself for an explicit stage)
    KE_i = fE(t_i, Y_i)
    u_n+1 = u_n + h sum_j (bt_j KI_j + b_j KE_j)
Stage solves: Newton + GMRES (ThetaStepper's core) or, with linear_solver="torch", a dense LU of
shift*I - J with J = d funcIM / du of ONE sample (``torch.func.jacrev`` on u[0:1], pa.py:474-481),
factored once per shift and applied to all batch rows with ``lu_solve(left=False)`` -- exact in
one Newton step when funcIM is linear and batch-row-wise (the Burgers/KS drivers' case).
Adjoint: see oracle/arkimex_oracle.py (the recurrence is restated there and checked against
autograd); transposed stage systems by GMRES on the transposed operator or the LU's adjoint solve.
Tableaus ``3`` (PETSc's default), ``4``, ``5``, ``l2``, ``ars122``, ``a2``, ``ars443``, ``prssp2``, ``bpr3``: coefficients
verified by ALL coupled order conditions up to the scheme's order (every bicoloured rooted tree,
tests/test_oracle_pins.py) -- including the four types examples-sinode/Burgers/run_a100_512.sh:20-23 selects.
``1bee`` (named in examples-sinode/Burgers/Burgers.py:19), ``2c``, ``2d``, ``2e``: restated from PETSc's manual pages
(what each scheme IS: backward Euler as two half steps with the full step embedded; the L-stable gamma = 1 - 1/sqrt 2
ESDIRK with three explicit companions) -- the implicit parts and every row sum are fixed by that description and by the
order conditions, the free entries of the explicit parts' last rows (2c: 1/2 1/2; 2d: 3/4 1/4; 2e: (3 -+ 2 sqrt 2)/6) are
from memory of PETSc's arkimex.c: PARITY UNPINNED for those six numbers (they do not change the order).
"""
from decimal import Decimal, getcontext
from fractions import Fraction as F

from . import _lib
from .theta import ThetaStepper

_g = F(1767732205903, 4055673282236)
_h = F(1, 2)
_q = F(1, 4)
_g5 = F(41, 200)
getcontext().prec = 60
_s2 = F(Decimal(2).sqrt())                   # sqrt(2) to 60 digits
_gl = 1 - 1 / _s2                            # 1 - 1/sqrt(2)
_t = F(1, 3)
# implicit part shared by PETSc's 2c / 2d / 2e: the stiffly accurate, L-stable 3-stage ESDIRK with gamma = 1 - 1/sqrt 2
# (TR-BDF2 written as a Runge-Kutta scheme); the three types differ in the explicit part's last row
_At2 = [[0, 0, 0], [_gl, _gl, 0], [1 / (2 * _s2), 1 / (2 * _s2), _gl]]
# name -> (order, A, At, b, bt or None)
TABLEAUS = {
    "3": (3,
          [[0, 0, 0, 0],
           [F(1767732205903, 2027836641118), 0, 0, 0],
           [F(5535828885825, 10492691773637), F(788022342437, 10882634858940), 0, 0],
           [F(6485989280629, 16251701735622), F(-4246266847089, 9704473918619), F(10755448449292, 10357097424841), 0]],
          [[0, 0, 0, 0],
           [_g, _g, 0, 0],
           [F(2746238789719, 10658868560708), F(-640167445237, 6845629431997), _g, 0],
           [F(1471266399579, 7840856788654), F(-4482444167858, 7529755066697), F(11266239266428, 11593286722821), _g]],
          [F(1471266399579, 7840856788654), F(-4482444167858, 7529755066697), F(11266239266428, 11593286722821), _g], None),
    # ARK4(3)6L[2]SA (Kennedy & Carpenter 2003): PETSc's TSARKIMEX4
    "4": (4,
          [[0, 0, 0, 0, 0, 0],
               [_h, 0, 0, 0, 0, 0],
               [F(13861, 62500), F(6889, 62500), 0, 0, 0, 0],
               [F(-116923316275, 2393684061468), F(-2731218467317, 15368042101831), F(9408046702089, 11113171139209), 0, 0, 0],
               [F(-451086348788, 2902428689909), F(-2682348792572, 7519795681897), F(12662868775082, 11960479115383),
                F(3355817975965, 11060851509271), 0, 0],
               [F(647845179188, 3216320057751), F(73281519250, 8382639484533), F(552539513391, 3454668386233),
                F(3354512671639, 8306763924573), F(4040, 17871), 0]],
          [[0, 0, 0, 0, 0, 0],
                [_q, _q, 0, 0, 0, 0],
                [F(8611, 62500), F(-1743, 31250), _q, 0, 0, 0],
                [F(5012029, 34652500), F(-654441, 2922500), F(174375, 388108), _q, 0, 0],
                [F(15267082809, 155376265600), F(-71443401, 120774400), F(730878875, 902184768), F(2285395, 8070912), _q, 0],
                [F(82889, 524892), 0, F(15625, 83664), F(69875, 102672), F(-2260, 8211), _q]],
          [F(82889, 524892), 0, F(15625, 83664), F(69875, 102672), F(-2260, 8211), _q], None),
    # ARK5(4)8L[2]SA (Kennedy & Carpenter 2003): PETSc's TSARKIMEX5
    "5": (5,
          [[0, 0, 0, 0, 0, 0, 0, 0],
               [F(41, 100), 0, 0, 0, 0, 0, 0, 0],
               [F(367902744464, 2072280473677), F(677623207551, 8224143866563), 0, 0, 0, 0, 0, 0],
               [F(1268023523408, 10340822734521), 0, F(1029933939417, 13636558850479), 0, 0, 0, 0, 0],
               [F(14463281900351, 6315353703477), 0, F(66114435211212, 5879490589093), F(-54053170152839, 4284798021562), 0, 0, 0, 0],
               [F(14090043504691, 34967701212078), 0, F(15191511035443, 11219624916014), F(-18461159152457, 12425892160975),
                F(-281667163811, 9011619295870), 0, 0, 0],
               [F(19230459214898, 13134317526959), 0, F(21275331358303, 2942455364971), F(-38145345988419, 4862620318723),
                F(-1, 8), F(-1, 8), 0, 0],
               [F(-19977161125411, 11928030595625), 0, F(-40795976796054, 6384907823539), F(177454434618887, 12078138498510),
                F(782672205425, 8267701900261), F(-69563011059811, 9646580694205), F(7356628210526, 4942186776405), 0]],
          [[0, 0, 0, 0, 0, 0, 0, 0],
                [_g5, _g5, 0, 0, 0, 0, 0, 0],
                [F(41, 400), F(-567603406766, 11931857230679), _g5, 0, 0, 0, 0, 0],
                [F(683785636431, 9252920307686), 0, F(-110385047103, 1367015193373), _g5, 0, 0, 0, 0],
                [F(3016520224154, 10081342136671), 0, F(30586259806659, 12414158314087), F(-22760509404356, 11113319521817), _g5, 0, 0, 0],
                [F(218866479029, 1489978393911), 0, F(638256894668, 5436446318841), F(-1179710474555, 5321154724896),
                 F(-60928119172, 8023461067671), _g5, 0, 0],
                [F(1020004230633, 5715676835656), 0, F(25762820946817, 25263940353407), F(-2161375909145, 9755907335909),
                 F(-211217309593, 5846859502534), F(-4269925059573, 7827059040749), _g5, 0],
                [F(-872700587467, 9133579230613), 0, 0, F(22348218063261, 9555858737531), F(-1143369518992, 8141816002931),
                 F(-39379526789629, 19018526304540), F(32727382324388, 42900044865799), _g5]],
          [F(-872700587467, 9133579230613), 0, 0, F(22348218063261, 9555858737531), F(-1143369518992, 8141816002931),
              F(-39379526789629, 19018526304540), F(32727382324388, 42900044865799), _g5], None),
    # Pareschi & Russo's SSP2(2,2,2): L-stable SDIRK pair, both stages implicit, c_E = [0,1] != c_I
    "l2": (2, [[0, 0], [1, 0]], [[_gl, 0], [1 - 2 * _gl, _gl]], [_h, _h], None),
    # --- PETSc types restated from its manual pages / the literature; see the module docstring for what pins each one
    # backward Euler taken as two half steps, the full step being the embedded solution ("extrapolation as error estimator")
    "1bee": (1, [[0, 0, 0], [0, 0, 0], [0, _h, 0]], [[1, 0, 0], [0, _h, 0], [0, _h, _h]], [0, _h, _h], None),
    "2c": (2, [[0, 0, 0], [2 - _s2, 0, 0], [_h, _h, 0]], _At2, _At2[2], None),
    "2d": (2, [[0, 0, 0], [2 - _s2, 0, 0], [F(3, 4), F(1, 4), 0]], _At2, _At2[2], None),
    "2e": (2, [[0, 0, 0], [2 - _s2, 0, 0], [(3 - 2 * _s2) / 6, (3 + 2 * _s2) / 6, 0]], _At2, _At2[2], None),
    # Pareschi & Russo 2005, SSP2(3,3,2)
    "prssp2": (2, [[0, 0, 0], [_h, 0, 0], [_h, _h, 0]], [[_q, 0, 0], [0, _q, 0], [_t, _t, _t]], [_t, _t, _t], None),
    # Boscarino, Pareschi & Russo 2013, BPR(3,5,3)
    "bpr3": (3,
             [[0, 0, 0, 0, 0], [1, 0, 0, 0, 0], [F(4, 9), F(2, 9), 0, 0, 0], [_q, 0, F(3, 4), 0, 0], [_q, 0, F(3, 4), 0, 0]],
             [[0, 0, 0, 0, 0], [_h, _h, 0, 0, 0], [F(5, 18), F(-1, 9), _h, 0, 0], [_h, 0, 0, _h, 0], [_q, 0, F(3, 4), -_h, _h]],
             [_q, 0, F(3, 4), -_h, _h], None),
    "ars122": (2, [[0, 0], [_h, 0]], [[0, 0], [0, _h]], [0, 1], None),
    "a2": (2, [[0, 0], [1, 0]], [[0, 0], [_h, _h]], [_h, _h], None),
    "ars443": (3,
               [[0, 0, 0, 0, 0], [_h, 0, 0, 0, 0], [F(11, 18), F(1, 18), 0, 0, 0], [F(5, 6), F(-5, 6), _h, 0, 0],
                [F(1, 4), F(7, 4), F(3, 4), F(-7, 4), 0]],
               [[0, 0, 0, 0, 0], [0, _h, 0, 0, 0], [0, F(1, 6), _h, 0, 0], [0, -_h, _h, _h, 0],
                [0, F(3, 2), F(-3, 2), _h, _h]],
               [F(1, 4), F(7, 4), F(3, 4), F(-7, 4), 0], [0, F(3, 2), F(-3, 2), _h, _h]),
}


# Embedded (order - 1) weights for the step-size controller, b^ = bt^ for these pairs.  ARK3(2)4L[2]SA, ARK4(3)6L[2]SA,
# ARK5(4)8L[2]SA: Kennedy & Carpenter 2003 (every coupled order condition up to order - 1 is checked in
# tests/test_oracle_pins.py); 1bee: the full backward-Euler step (stage 0).
EMBEDDED = {
    "3": [F(2756255671327, 12835298489170), F(-10771552573575, 22201958757719), F(9247589265047, 10645013368117),
          F(2193209047091, 5459859503100)],
    "4": [F(4586570599, 29645900160), 0, F(178811875, 945068544), F(814220225, 1159782912), F(-3700637, 11593932),
          F(61727, 225920)],
    "5": [F(-975461918565, 9796059967033), 0, 0, F(78070527104295, 32432590147079), F(-548382580838, 3424219808633),
          F(-33438840321285, 15594753105479), F(3629800801594, 4656183773603), F(4035322873751, 18575991585200)],
    "1bee": [1, 0, 0],
}


# The order PETSc REGISTERS a type with is the exponent TSAdaptChoose_Basic uses (h_new = h * safety * e^(-1/order)).  It is
# the accuracy order except for 1bee: TSARKIMEXRegister(TSARKIMEX1BEE, 2, 3, ...) -- backward Euler with the extrapolated
# pair is registered as order 2.
CONTROLLER_ORDER = {"1bee": 2}


def get_tableau(name):
    if name not in TABLEAUS:
        raise _lib.PnError("ARKIMEX type %r is not available (have: %s)" % (name, ", ".join(sorted(TABLEAUS))))
    order, A, At, b, bt = TABLEAUS[name]
    A = [[float(x) for x in r] for r in A]
    At = [[float(x) for x in r] for r in At]
    b = [float(x) for x in b]
    bt = b if bt is None else [float(x) for x in bt]
    # PETSc evaluates the implicit part at t + ct_i h and the explicit part at t + c_i h (row sums)
    return dict(s=len(b), order=order, adapt_order=CONTROLLER_ORDER.get(name, order), A=A, At=At, b=b, bt=bt,
                c=[sum(r) for r in At], cE=[sum(r) for r in A])


class ArkimexStepper(ThetaStepper):
    tapes_ex = True          # the explicitly treated part's stage evaluations can be kept as autograd tapes (theta.py::odeint)

    def __init__(self, ode, db):
        ThetaStepper.__init__(self, ode, "beuler", db)       # Newton/GMRES options and buffers
        self.method = "imex"
        self.which = "IM"
        self.name = str(db.get("ts_arkimex_type", "3"))
        self.tab = get_tableau(self.name)
        self._K = None
        if ode.mass is not None:
            raise NotImplementedError("IMEX with a mass matrix is not built")

    # ---------------------------------------------------------------- helpers
    def _lincomb_many(self, out, xs, cs):
        """out = sum c_j x_j for any number of terms (chunks of 8 through the streaming kernel)."""
        ops = self.ode._ops
        terms = [(x, c) for x, c in zip(xs, cs) if c != 0.0]
        first = terms[:8]
        ops.lincomb(out, [x for x, _ in first], [c for _, c in first])
        k = 8
        while k < len(terms):
            chunk = terms[k:k + 7]
            ops.lincomb(out, [out] + [x for x, _ in chunk], [1.0] + [c for _, c in chunk])
            k += 7

    def _implicit_stages(self):
        return sum(1 for i in range(self.tab["s"]) if self.tab["At"][i][i] != 0.0)

    # ---------------------------------------------------------------- one step
    def nstage(self):
        return self.tab["s"]

    def _do_step(self, tn, h, u, unew, stage_dest):
        """One ARK-IMEX step u -> unew; stage value Y_i is written to stage_dest(i).  Returns [Y_i]."""
        o, ops, tab = self.ode, self.ode._ops, self.tab
        s, A, At, b, bt, c = tab["s"], tab["A"], tab["At"], tab["b"], tab["bt"], tab["c"]
        Y, KI, KE = [], [], []
        for i in range(s):
            ti = tn + c[i] * h
            Z = self._buf("Z")
            xs, cs = [u], [1.0]
            for j in range(i):
                xs += [KI[j], KE[j]]
                cs += [h * At[i][j], h * A[i][j]]
            self._lincomb_many(Z, xs, cs)
            y = stage_dest(i)
            if At[i][i] != 0.0:
                # Newton's initial guess: the previous stage value (the state for the first stage), as
                # TSStep_ARKIMEX takes it without -ts_arkimex_initial_guess_extrapolate.  It decides the
                # result when the solve is cut short (-snes_type ksponly) and funcIM is nonlinear.
                ops.copy(y, Y[i - 1] if i > 0 else u)
                shift = 1.0 / (h * At[i][i])
                lin = self._direct_solver(ti, u, shift, False) if self.direct else None
                self._newton(ti, shift, Z, None, y, lin)
                ki = self._buf("KI%d" % i)
                ops.lincomb(ki, [y, Z], [shift, -shift])
            else:
                ops.copy(y, Z)
                ki = self._f(ti, y, "IM")
            Y.append(y)
            KI.append(ki)
            if self._tape_rec is not None:
                rec = []
                KE.append(o._call_func(tn + tab["cE"][i] * h, y, rec))      # recorded by autograd: (input, output, parameters)
                self._tape_rec[i] = rec[0]
            else:
                KE.append(self._f(tn + tab["cE"][i] * h, y, "EX"))
        xs, cs = [u], [1.0]
        for j in range(s):
            xs += [KI[j], KE[j]]
            cs += [h * bt[j], h * b[j]]
        self._lincomb_many(unew, xs, cs)
        self._K = (KI, KE)
        return Y

    def embedded(self):
        """Embedded weights b^ (= bt^) as floats, or None when the type has none here."""
        e = EMBEDDED.get(self.name)
        return None if e is None else [float(x) for x in e]

    def lte_available(self):
        return False

    def error_begin(self):
        pass

    def error_accept(self, h, u):
        pass

    def error_norm(self, h, u, unew):
        """TSEvaluateStep_ARKIMEX(order - 1) + TSErrorWeightedNorm for the step just taken: the embedded solution is
        unew + h sum_j (b^_j - bt_j) KI_j + (b^_j - b_j) KE_j; its WRMS distance to unew goes to the pinned scalar."""
        o, ops, tab = self.ode, self.ode._ops, self.tab
        be = self.embedded()
        KI, KE = self._K
        xs, cs = [], []
        for j in range(tab["s"]):
            xs += [KI[j], KE[j]]
            cs += [h * (be[j] - tab["bt"][j]), h * (be[j] - tab["b"][j])]
        E = self._buf("E")
        if any(c != 0.0 for c in cs):
            self._lincomb_many(E, xs, cs)
        else:
            ops.lincomb(E, [unew], [0.0])
        ops.combine_wrms(None, unew, [E], [0.0], [1.0], o._atol, o._rtol)
        return True

    # ---------------------------------------------------------------- reverse sweep
    def adjoint_steps(self, nsteps, forcing):
        o, ops, tab = self.ode, self.ode._ops, self.tab
        s, A, At, b, bt, c = tab["s"], tab["A"], tab["At"], tab["b"], tab["bt"], tab["c"]
        lam = o.adj_u_flat
        for r in range(nsteps):
            step = o._rev_next
            tn, h = o._step_info(step)
            u, Y = self._stages_of(step)
            tapes = o._tapes.pop(step, None) if o._tapes else None
            nu = [None] * s
            for i in range(s - 1, -1, -1):
                ti = tn + c[i] * h
                # cotangents of the stage derivatives
                keb, kib = self._buf("KEb"), self._buf("KIb")
                xe, ce, xi, ci = [lam], [h * b[i]], [lam], [h * bt[i]]
                for k in range(i + 1, s):
                    xe.append(nu[k]); ce.append(h * A[k][i])
                    xi.append(nu[k]); ci.append(h * At[k][i])
                use_e = any(cc != 0.0 for cc in ce)
                use_i = any(cc != 0.0 for cc in ci)
                ybar = self._buf("ybar")
                terms = []
                if use_e:
                    self._lincomb_many(keb, xe, ce)
                    gE, gpE = o._vjp(tn + tab["cE"][i] * h, Y[i], keb, tapes[i] if tapes else None, which="EX", alpha=1.0)
                    if tapes:
                        tapes[i] = None                    # release the stage's activations as soon as they are used
                    if gE is not None:
                        terms.append(gE)
                    if o.npEX > 0:
                        o._add_param_grads(1.0, gpE, first=len(o._poffI))
                if use_i:
                    self._lincomb_many(kib, xi, ci)
                    if self._affine:                       # funcIM is affine with the kept Jacobian (theta.py::_check_affine)
                        gI, gpI = self._jt_rows(kib), []
                    else:
                        gI, gpI = o._vjp(ti, Y[i], kib, which="IM")
                    if gI is not None:
                        terms.append(gI)
                    if o.npIM > 0:
                        o._add_param_grads(1.0, gpI, first=0)
                nui = self._buf("nu%d" % i)
                if not terms:
                    ops.lincomb(nui, [lam], [0.0])
                    nu[i] = nui
                    continue
                ops.lincomb(ybar, terms, [1.0] * len(terms))
                if At[i][i] != 0.0:
                    hg = h * At[i][i]
                    shift = 1.0 / hg
                    # (I - hg J)^T nu = ybar  <=>  (shift I - J)^T nu = shift ybar
                    ops.lincomb(ybar, [ybar], [shift])
                    stable = True
                    if self.direct:
                        self._direct_solver(ti, u, shift, True)(ybar, nui)
                        gp2 = o._vjp(ti, Y[i], nui, which="IM")[1] if o.npIM > 0 else []
                    else:
                        ent = self._op_graph(ti, True, Y[i])
                        if ent is not None:                      # replayed linearisation of funcIM at this stage time
                            ent.linearise(Y[i])
                            self._gmres(None, shift, ybar, nui, True, graph=ent)
                            gp2 = ent.param_cotangents(nui, self._kr) if o.npIM > 0 else []
                            stable = False                       # static outputs of the replayed graph
                        else:
                            jt, _ = self._linearise(ti, Y[i], True)
                            self._gmres(jt, shift, ybar, nui, True)
                            gp2 = o._vjp(ti, Y[i], nui, which="IM")[1] if o.npIM > 0 else []
                    if o.npIM > 0:
                        o._add_param_grads(hg, gp2, first=0, stable=stable)
                else:
                    ops.copy(nui, ybar)
                nu[i] = nui
            if o._pend_g and (o._accum_mode == "step" or len(o._pend_g) + 3 * s > o._accum_cap):
                o._flush_param_accum()           # mu += the queued stage results, oldest first: one launch
            elif o._pend_bias and o._accum_mode == "step":
                o._flush_bias_accum()
            self._lincomb_many(lam, [lam] + nu, [1.0] * (s + 1))
            o._traj.rev_done(step)
            o._rev_next = step - 1
        if forcing is not None:
            ops.lincomb(lam, [lam, forcing], [1.0, 1.0])
