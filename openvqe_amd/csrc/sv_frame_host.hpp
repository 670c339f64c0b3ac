// sv_frame_host.hpp — host-side compiler pieces of the Clifford-frame form of a literal gate list (ovqe_set_gate_program,
// ref:openvqe/common_files/circuit.py:13-106: the QUCCSD templates are Clifford gates around RZ / RY rotations): pure C++17, no HIP,
// so that tests/cpu/clifford_frame_check.cpp can run them under AddressSanitizer / UBSan against a dense simulation of the gates.
//
//   U = G_L ... G_1,  Clifford gates folded into a frame C_k = (Clifford gates up to k):  a rotation exp(-i phi P) behind C_k becomes
//   C_k^+ exp(-i phi P) C_k = exp(-i phi (C_k^+ P C_k)), so U = C_total * prod_k exp(-i phi_k P'_k) with every P'_k one Pauli string.
//   The frame is the images of X_q and Z_q under conjugation (2 n Pauli strings with phases), updated per gate in O(1) words.
#pragma once
#include <cmath>
#include <complex>
#include <cstdint>
#include <unordered_map>
#include <vector>

#ifndef OVQE_GATE_X
#define OVQE_GATE_X 0
#define OVQE_GATE_H 1
#define OVQE_GATE_RX 2
#define OVQE_GATE_RY 3
#define OVQE_GATE_RZ 4
#define OVQE_GATE_CNOT 5
#endif

namespace ovqe_frame {

struct PauliRaw {  // i^k X^x Z^z
    uint64_t x, z;
    int k;
};
inline PauliRaw pauli_mul(const PauliRaw &a, const PauliRaw &b) {
    return PauliRaw{a.x ^ b.x, a.z ^ b.z, (a.k + b.k + 2 * __builtin_popcountll(a.z & b.x)) & 3};
}

struct FrameEmit {   // one rotation exp(-i (coeff theta_pidx + phi0) P), P = i^{|x&z|} X^x Z^z, in frame form
    uint64_t x, z;
    double coeff, phi0;
    int32_t pidx;
};
struct FrameTrack {
    std::vector<PauliRaw> ix, iz;      // images of X_q, Z_q after the whole list
    std::vector<FrameEmit> emitted;    // the rotations, in order
    std::vector<int64_t> tail;         // indices of the gates folded into the frame (the Clifford part, original order)
    bool closed = false;               // the frame is the identity again: C_total is a global phase
};

// gate g: opcode[g] on bit b0[g] (CNOT: control b0, target b1), rotation angle = ascale[g] * theta[pidx[g]] + aconst[g] (pidx < 0: constant);
// X, H, CNOT and constant quarter turns (|angle| = pi/2) go into the frame, every other rotation is emitted.
// false: a conjugated generator came out non-Hermitian (cannot happen for a consistent frame: internal error)
inline bool track_clifford_frame(int n, int64_t G, const int32_t *opcode, const int32_t *b0, const int32_t *b1, const double *ascale,
                                 const double *aconst, const int32_t *pidx, FrameTrack &F) {
    F.ix.resize(n);
    F.iz.resize(n);
    F.emitted.clear();
    F.tail.clear();
    std::vector<PauliRaw> &ix = F.ix, &iz = F.iz;
    for (int q = 0; q < n; ++q) {
        ix[q] = PauliRaw{1ull << q, 0, 0};
        iz[q] = PauliRaw{0, 1ull << q, 0};
    }
    for (int64_t g = 0; g < G; ++g) {
        const int t = b0[g];
        switch (opcode[g]) {
        case OVQE_GATE_H: std::swap(ix[t], iz[t]); F.tail.push_back(g); continue;
        case OVQE_GATE_X: iz[t].k = (iz[t].k + 2) & 3; F.tail.push_back(g); continue;
        case OVQE_GATE_CNOT:
            ix[t] = pauli_mul(ix[t], ix[b1[g]]);          // X_c -> X_c X_t
            iz[b1[g]] = pauli_mul(iz[t], iz[b1[g]]);      // Z_t -> Z_c Z_t
            F.tail.push_back(g);
            continue;
        default: break;
        }
        const double phi0 = 0.5 * aconst[g], coeff = 0.5 * ascale[g];
        if (pidx[g] < 0 && std::fabs(std::fabs(phi0) - M_PI_4) < 1e-15) {
            // quarter turn exp(-i s pi/4 P): Q -> i s P Q for the generators anticommuting with P
            const int s = phi0 > 0 ? 1 : 3;  // i^1 = i, i^3 = -i
            const PauliRaw X = ix[t], Z = iz[t];
            if (opcode[g] == OVQE_GATE_RZ) {
                ix[t] = pauli_mul(PauliRaw{0, 0, s}, pauli_mul(Z, X));
            } else if (opcode[g] == OVQE_GATE_RX) {
                iz[t] = pauli_mul(PauliRaw{0, 0, s}, pauli_mul(X, Z));
            } else {  // RY: X -> s Z, Z -> -s X
                ix[t] = pauli_mul(PauliRaw{0, 0, s == 1 ? 0 : 2}, Z);
                iz[t] = pauli_mul(PauliRaw{0, 0, s == 1 ? 2 : 0}, X);
            }
            F.tail.push_back(g);
            continue;
        }
        PauliRaw P;
        if (opcode[g] == OVQE_GATE_RX) P = ix[t];
        else if (opcode[g] == OVQE_GATE_RZ) P = iz[t];
        else P = pauli_mul(PauliRaw{0, 0, 1}, pauli_mul(ix[t], iz[t]));  // Y = i X Z
        const int rel = (P.k - __builtin_popcountll(P.x & P.z)) & 3;    // Hermitian string = i^{|x&z|} X^x Z^z
        if (rel & 1) return false;
        const double sg = rel ? -1.0 : 1.0;
        F.emitted.push_back(FrameEmit{P.x, P.z, sg * coeff, sg * phi0, pidx[g]});
    }
    F.closed = true;
    for (int q = 0; q < n && F.closed; ++q)
        F.closed = ix[q].x == (1ull << q) && ix[q].z == 0 && ix[q].k == 0 && iz[q].x == 0 && iz[q].z == (1ull << q) && iz[q].k == 0;
    return true;
}

// <hf| C |hf> of the Clifford part C of a gate list (the gates `tail`: X, H, CNOT, quarter turns) by a SPARSE simulation on the host:
// between the basis changes of one excitation template and their inverses the state is a superposition of a handful of basis states,
// so the 49 272 Clifford gates of the N2 QUCCSD list cost a few milliseconds here against 0.4 s as a literal program on the 2^24
// register (which was most of ovqe_set_gate_program's time).  false: more than `cap` basis states at some point — the caller runs
// the gates on the device instead.
inline bool clifford_amplitude_on_host(uint64_t hf, const std::vector<int64_t> &tail, const int32_t *opcode, const int32_t *b0,
                                       const int32_t *b1, const double *aconst, std::complex<double> *amp, size_t cap = 4096) {
    using cd = std::complex<double>;
    std::unordered_map<uint64_t, cd> cur, nxt;
    cur.emplace(hf, cd(1.0, 0.0));
    const double r = 0.70710678118654752440;
    for (const int64_t g : tail) {
        const uint64_t bt = 1ull << b0[g];
        const int op = opcode[g];
        if (op == OVQE_GATE_X || op == OVQE_GATE_CNOT) {   // permutations
            nxt.clear();
            const uint64_t flip = op == OVQE_GATE_X ? bt : (1ull << b1[g]);
            for (const auto &kv : cur) nxt.emplace((op == OVQE_GATE_X || (kv.first & bt)) ? kv.first ^ flip : kv.first, kv.second);
            cur.swap(nxt);
            continue;
        }
        if (op == OVQE_GATE_RZ) {   // exp(-i phi Z), phi = aconst / 2 = +- pi/4: diagonal
            const double sg = aconst[g] > 0 ? 1.0 : -1.0;
            for (auto &kv : cur) kv.second *= (kv.first & bt) ? cd(r, sg * r) : cd(r, -sg * r);
            continue;
        }
        // H, RX, RY: |b> -> u_bb |b> + u_{1-b,b} |1-b>
        cd u[2][2];   // u[row][column]
        if (op == OVQE_GATE_H) {
            u[0][0] = r; u[0][1] = r; u[1][0] = r; u[1][1] = -r;
        } else {
            const double sg = aconst[g] > 0 ? 1.0 : -1.0;
            if (op == OVQE_GATE_RX) {        // cos - i sin X
                u[0][0] = r; u[1][1] = r; u[0][1] = cd(0.0, -sg * r); u[1][0] = cd(0.0, -sg * r);
            } else {                         // RY: cos - i sin Y,  Y = [[0, -i], [i, 0]]
                u[0][0] = r; u[1][1] = r; u[0][1] = -sg * r; u[1][0] = sg * r;
            }
        }
        nxt.clear();
        for (const auto &kv : cur) {
            const int bit = (kv.first & bt) ? 1 : 0;
            nxt[kv.first] += u[bit][bit] * kv.second;
            nxt[kv.first ^ bt] += u[1 - bit][bit] * kv.second;
        }
        cur.clear();
        for (const auto &kv : nxt)
            if (std::abs(kv.second) > 1e-13) cur.emplace(kv.first, kv.second);   // what the inverse basis change cancels
        if (cur.size() > cap) return false;
    }
    const auto it = cur.find(hf);
    *amp = it == cur.end() ? cd(0.0, 0.0) : it->second;
    return true;
}

}  // namespace ovqe_frame
