// CPU self-check of the Clifford-frame compiler pieces (openvqe_amd/csrc/sv_frame_host.hpp), compiled with g++ under
// AddressSanitizer + UBSan by tests/test_sanitizer.py.
//
// Random gate lists (X, H, CNOT, RX / RY / RZ with constant quarter turns and with parametrised angles) on up to 7 qubits:
//   (1) the frame form — the emitted Pauli rotations applied to |hf>, then the Clifford part (the `tail` gates) — must give the state
//       of the literal list, amplitude by amplitude (dense simulation here, gate by gate);
//   (2) <hf| C |hf> of the Clifford part from the sparse host simulation must equal the dense one;
//   (3) the sparse simulation gives up (returns false) when it exceeds its cap.
// usage: clifford_frame_check [cases] [seed]
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../openvqe_amd/csrc/sv_frame_host.hpp"

using cd = std::complex<double>;
using namespace ovqe_frame;

static void apply_1q(std::vector<cd> &psi, int bit, const cd u[2][2]) {
    const uint64_t m = 1ull << bit;
    for (uint64_t i = 0; i < psi.size(); ++i)
        if (!(i & m)) {
            const cd a = psi[i], b = psi[i | m];
            psi[i] = u[0][0] * a + u[0][1] * b;
            psi[i | m] = u[1][0] * a + u[1][1] * b;
        }
}
static void apply_gate(std::vector<cd> &psi, int op, int t, int t2, double angle) {
    const double c = std::cos(0.5 * angle), s = std::sin(0.5 * angle), r = std::sqrt(0.5);
    cd u[2][2];
    switch (op) {
    case OVQE_GATE_X: u[0][0] = 0; u[0][1] = 1; u[1][0] = 1; u[1][1] = 0; apply_1q(psi, t, u); return;
    case OVQE_GATE_H: u[0][0] = r; u[0][1] = r; u[1][0] = r; u[1][1] = -r; apply_1q(psi, t, u); return;
    case OVQE_GATE_RX: u[0][0] = c; u[1][1] = c; u[0][1] = cd(0, -s); u[1][0] = cd(0, -s); apply_1q(psi, t, u); return;
    case OVQE_GATE_RY: u[0][0] = c; u[1][1] = c; u[0][1] = -s; u[1][0] = s; apply_1q(psi, t, u); return;
    case OVQE_GATE_RZ: u[0][0] = cd(c, -s); u[1][1] = cd(c, s); u[0][1] = 0; u[1][0] = 0; apply_1q(psi, t, u); return;
    case OVQE_GATE_CNOT: {
        const uint64_t mc = 1ull << t, mt = 1ull << t2;
        for (uint64_t i = 0; i < psi.size(); ++i)
            if ((i & mc) && !(i & mt)) std::swap(psi[i], psi[i | mt]);
        return;
    }
    }
}
// exp(-i phi P), P = i^{|x&z|} X^x Z^z:  P|i> = i^{|x&z|} (-1)^{|i&z|} |i ^ x>
static void apply_rotation(std::vector<cd> &psi, uint64_t x, uint64_t z, double phi) {
    const cd iy[4] = {cd(1, 0), cd(0, 1), cd(-1, 0), cd(0, -1)};
    const cd ph = iy[__builtin_popcountll(x & z) & 3];
    std::vector<cd> out(psi.size());
    const double c = std::cos(phi), s = std::sin(phi);
    for (uint64_t i = 0; i < psi.size(); ++i) {
        const cd pi = ph * ((__builtin_popcountll(i & z) & 1) ? -1.0 : 1.0) * psi[i];   // coefficient of |i ^ x> in P psi
        out[i] += c * psi[i];
        out[i ^ x] += cd(0, -s) * pi;
    }
    psi.swap(out);
}

int main(int argc, char **argv) {
    const int cases = argc > 1 ? std::atoi(argv[1]) : 300;
    std::mt19937_64 rng(argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 1);
    auto rnd = [&](int n) { return (int)(rng() % (uint64_t)n); };
    std::uniform_real_distribution<double> ang(-2.0, 2.0);
    double worst_state = 0.0, worst_amp = 0.0;
    int closed_cases = 0, rotations = 0;
    for (int c = 0; c < cases; ++c) {
        const int n = 2 + rnd(6), G = 1 + rnd(60), K = 1 + rnd(4);
        std::vector<int32_t> opcode(G), b0(G), b1(G), pidx(G);
        std::vector<double> ascale(G), aconst(G), theta(K);
        for (double &t : theta) t = ang(rng);
        const bool make_closed = c % 3 == 0;   // second half = inverse Clifford gates: the frame closes
        const int half = make_closed ? G / 2 : G;
        for (int g = 0; g < half; ++g) {
            const int kind = rnd(make_closed ? 6 : 9);
            b0[g] = rnd(n); b1[g] = 0; pidx[g] = -1; ascale[g] = 0.0; aconst[g] = 0.0;
            if (kind == 0) opcode[g] = OVQE_GATE_X;
            else if (kind == 1) opcode[g] = OVQE_GATE_H;
            else if (kind == 2 && n > 1) { opcode[g] = OVQE_GATE_CNOT; do b1[g] = rnd(n); while (b1[g] == b0[g]); }
            else if (kind <= 5) { opcode[g] = OVQE_GATE_RX + rnd(3); aconst[g] = (rnd(2) ? 1.0 : -1.0) * M_PI_2; }   // quarter turn: Clifford
            else { opcode[g] = OVQE_GATE_RX + rnd(3); pidx[g] = rnd(3) ? rnd(K) : -1; ascale[g] = pidx[g] < 0 ? 0.0 : ang(rng); aconst[g] = ang(rng); }
        }
        if (make_closed) {
            for (int g = half; g < 2 * half; ++g) {   // inverse of gate (2 half - 1 - g)
                const int s = 2 * half - 1 - g;
                opcode[g] = opcode[s]; b0[g] = b0[s]; b1[g] = b1[s]; pidx[g] = -1; ascale[g] = 0.0; aconst[g] = -aconst[s];
            }
            for (int g = 2 * half; g < G; ++g) { opcode[g] = OVQE_GATE_RZ; b0[g] = rnd(n); b1[g] = 0; pidx[g] = rnd(K); ascale[g] = ang(rng); aconst[g] = ang(rng); }
        }
        const uint64_t hf = rng() & ((1ull << n) - 1ull);
        // literal
        std::vector<cd> lit((size_t)1 << n, cd(0, 0));
        lit[hf] = 1.0;
        for (int g = 0; g < G; ++g) apply_gate(lit, opcode[g], b0[g], b1[g], (pidx[g] < 0 ? 0.0 : ascale[g] * theta[pidx[g]]) + aconst[g]);
        // frame form
        FrameTrack F;
        if (!track_clifford_frame(n, G, opcode.data(), b0.data(), b1.data(), ascale.data(), aconst.data(), pidx.data(), F)) {
            std::printf("case %d: non-Hermitian generator\n", c);
            return 1;
        }
        std::vector<cd> fr((size_t)1 << n, cd(0, 0));
        fr[hf] = 1.0;
        for (const FrameEmit &e : F.emitted) apply_rotation(fr, e.x, e.z, (e.pidx < 0 ? 0.0 : e.coeff * theta[e.pidx]) + e.phi0);
        rotations += (int)F.emitted.size();
        std::vector<cd> cl((size_t)1 << n, cd(0, 0));   // the Clifford part alone on |hf>
        cl[hf] = 1.0;
        for (const int64_t g : F.tail) {
            apply_gate(fr, opcode[g], b0[g], b1[g], aconst[g]);
            apply_gate(cl, opcode[g], b0[g], b1[g], aconst[g]);
        }
        for (size_t i = 0; i < lit.size(); ++i) worst_state = std::max(worst_state, std::abs(lit[i] - fr[i]));
        if (worst_state > 1e-11) {
            std::printf("case %d (n = %d, %d gates): frame form differs from the literal list by %.3e\n", c, n, G, worst_state);
            return 1;
        }
        if (F.closed) {
            ++closed_cases;
            if (std::abs(std::abs(cl[hf]) - 1.0) > 1e-11) {
                std::printf("case %d: closed frame but |<hf|C|hf>| = %.15f\n", c, std::abs(cl[hf]));
                return 1;
            }
        }
        std::complex<double> amp;
        if (!clifford_amplitude_on_host(hf, F.tail, opcode.data(), b0.data(), b1.data(), aconst.data(), &amp, (size_t)1 << n)) {
            std::printf("case %d: sparse simulation gave up below its cap\n", c);
            return 1;
        }
        worst_amp = std::max(worst_amp, std::abs(amp - cl[hf]));
        if (worst_amp > 1e-11) {
            std::printf("case %d: sparse <hf|C|hf> differs from the dense one by %.3e\n", c, worst_amp);
            return 1;
        }
    }
    {   // the cap: n Hadamards spread |0> over 2^n basis states
        const int n = 6;
        std::vector<int32_t> opcode(n, OVQE_GATE_H), b0(n), b1(n, 0);
        std::vector<double> aconst(n, 0.0);
        std::vector<int64_t> tail(n);
        for (int q = 0; q < n; ++q) { b0[q] = q; tail[q] = q; }
        std::complex<double> amp;
        if (clifford_amplitude_on_host(0, tail, opcode.data(), b0.data(), b1.data(), aconst.data(), &amp, 16)) {
            std::printf("the sparse simulation did not give up at its cap\n");
            return 1;
        }
        if (!clifford_amplitude_on_host(0, tail, opcode.data(), b0.data(), b1.data(), aconst.data(), &amp, 64) || std::abs(amp - 0.125) > 1e-14) {
            std::printf("six Hadamards: amplitude %.15f\n", amp.real());
            return 1;
        }
    }
    std::printf("clifford frame ok: %d cases (%d closed frames, %d emitted rotations), worst |dpsi| %.2e, worst |damp| %.2e\n", cases, closed_cases,
                rotations, worst_state, worst_amp);
    return 0;
}
