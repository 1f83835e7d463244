// kinematics_kernels.hip - rigid-body kinematics of a heliostat field in one launch (gfx950).
//
// Replaces RigidBody.incident_ray_directions_to_orientations / motor_positions_to_orientations
// (artist/field/kinematics_rigid_body.py:194-634) with the ideal and linear actuator models
// (artist/field/actuators_ideal.py:87-111, actuators_linear.py:105-370).  The reference evaluates them as ~60 small
// batched ATen ops per iteration ([H,4,4] matmuls, trigonometry, clamps); here one thread owns one heliostat and
// keeps its 4x4 chain in registers.  The only coupling between heliostats is the stopping rule of the iterative
// alignment - "stop when NO heliostat's loss moved by more than min_eps" (:618-626): one launch per evaluation,
// the field-wide answer handed from launch to launch through a flag in HBM (see rigid_body_eval_kernel).
//
// Backward = forward-mode differentiation: the same templated code runs on dual numbers (value, derivative), one
// thread per (heliostat, parameter) seeds its parameter and contracts d(orientation) with dL/d(orientation).  17
// parameters per heliostat (4 rotation deviations, 9 translation deviations, 2 x 2 optimisable actuator
// parameters) x a few hundred flops: nothing next to the trace, and no hand-derived adjoint to get wrong.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "launch_common.hpp"

namespace art {

struct Dual {            // first-order dual number
    float v, d;
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) { const float q = a.v / b.v; return {q, (a.d - q * b.d) / b.v}; }

template <typename T> struct Num;
template <> struct Num<float> {
    static __device__ __forceinline__ float lift(float x) { return x; }
    static __device__ __forceinline__ float val(float x) { return x; }
    static __device__ __forceinline__ float sin_(float x) { return sinf(x); }
    static __device__ __forceinline__ float cos_(float x) { return cosf(x); }
    static __device__ __forceinline__ float sqrt_(float x) { return sqrtf(x); }
    static __device__ __forceinline__ float asin_(float x) { return asinf(x); }
    static __device__ __forceinline__ float acos_(float x) { return acosf(x); }
    static __device__ __forceinline__ float atan2_(float y, float x) { return atan2f(y, x); }
    static __device__ __forceinline__ float abs_(float x) { return fabsf(x); }
    static __device__ __forceinline__ float clamp_(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
    static __device__ __forceinline__ float softplus100(float x) { const float b = 100.0f * x; return b > 20.0f ? x : logf(1.0f + expf(b)) / 100.0f; }
};
template <> struct Num<Dual> {
    static __device__ __forceinline__ Dual lift(float x) { return {x, 0.0f}; }
    static __device__ __forceinline__ float val(Dual x) { return x.v; }
    static __device__ __forceinline__ Dual sin_(Dual x) { return {sinf(x.v), cosf(x.v) * x.d}; }
    static __device__ __forceinline__ Dual cos_(Dual x) { return {cosf(x.v), -sinf(x.v) * x.d}; }
    static __device__ __forceinline__ Dual sqrt_(Dual x) { const float s = sqrtf(x.v); return {s, x.d / (2.0f * s)}; }
    static __device__ __forceinline__ Dual asin_(Dual x) { return {asinf(x.v), x.d / sqrtf(1.0f - x.v * x.v)}; }
    static __device__ __forceinline__ Dual acos_(Dual x) { return {acosf(x.v), -x.d / sqrtf(1.0f - x.v * x.v)}; }
    static __device__ __forceinline__ Dual atan2_(Dual y, Dual x) { const float r2 = x.v * x.v + y.v * y.v; return {atan2f(y.v, x.v), (x.v * y.d - y.v * x.d) / r2}; }
    static __device__ __forceinline__ Dual abs_(Dual x) { return x.v < 0.0f ? Dual{-x.v, -x.d} : (x.v > 0.0f ? x : Dual{0.0f, 0.0f}); }
    // torch.clamp passes the gradient where min <= x <= max
    static __device__ __forceinline__ Dual clamp_(Dual x, float lo, float hi) { return x.v < lo ? Dual{lo, 0.0f} : (x.v > hi ? Dual{hi, 0.0f} : x); }
    static __device__ __forceinline__ Dual softplus100(Dual x)
    {
        const float b = 100.0f * x.v;
        if (b > 20.0f) return x;
        const float e = expf(b);
        return {logf(1.0f + e) / 100.0f, x.d * e / (1.0f + e)};
    }
};

template <typename T> struct Mat4 { T m[16]; };

template <typename T> __device__ __forceinline__ Mat4<T> m_eye()
{
    Mat4<T> r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r.m[i] = Num<T>::lift(i % 5 == 0 ? 1.0f : 0.0f);
    return r;
}
template <typename T> __device__ __forceinline__ Mat4<T> m_mul(const Mat4<T>& a, const Mat4<T>& b)
{
    Mat4<T> r;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            T acc = a.m[4 * i] * b.m[j];
#pragma unroll
            for (int k = 1; k < 4; ++k) acc = acc + a.m[4 * i + k] * b.m[4 * k + j];
            r.m[4 * i + j] = acc;
        }
    return r;
}
// artist/geometry/transforms.py:86-262
template <typename T> __device__ __forceinline__ Mat4<T> m_rot_e(T x) { Mat4<T> r = m_eye<T>(); const T c = Num<T>::cos_(x), s = Num<T>::sin_(x); r.m[5] = c; r.m[6] = -s; r.m[9] = s; r.m[10] = c; return r; }
template <typename T> __device__ __forceinline__ Mat4<T> m_rot_n(T x) { Mat4<T> r = m_eye<T>(); const T c = Num<T>::cos_(x), s = Num<T>::sin_(x); r.m[0] = c; r.m[2] = -s; r.m[8] = s; r.m[10] = c; return r; }
template <typename T> __device__ __forceinline__ Mat4<T> m_rot_u(T x) { Mat4<T> r = m_eye<T>(); const T c = Num<T>::cos_(x), s = Num<T>::sin_(x); r.m[0] = c; r.m[1] = -s; r.m[4] = s; r.m[5] = c; return r; }
template <typename T> __device__ __forceinline__ Mat4<T> m_trans(T e, T n, T u) { Mat4<T> r = m_eye<T>(); r.m[3] = e; r.m[7] = n; r.m[11] = u; return r; }

template <typename T> struct Actuator {        // one column of the actuator tables
    bool linear, clockwise;
    float min_pos, max_pos;
    float increment, offset, pivot;            // softplus(.) + 1e-6 of the non-optimisable rows (never differentiated)
    T init_angle, init_stroke;
};

template <typename T> struct Params {          // everything of one heliostat that can carry a derivative
    float pos[3];
    T rot[4], tr[9];
    Actuator<T> a1, a2;
};

// actuators_linear.py:206-233
template <typename T> __device__ __forceinline__ T act_abs_angle(const Actuator<T>& a, T motor)
{
    using N = Num<T>;
    T stroke = motor / N::lift(a.increment) + a.init_stroke;
    stroke = N::clamp_(stroke, fabsf(a.offset - a.pivot) + 1e-6f, a.offset + a.pivot - 1e-6f);
    const T num = N::lift(a.offset * a.offset + a.pivot * a.pivot) - stroke * stroke;
    return N::acos_(N::clamp_(num / N::lift(2.0f * a.offset * a.pivot), -1.0f + 1e-6f, 1.0f - 1e-6f));
}
// ideal :87 / linear :260-291
template <typename T> __device__ __forceinline__ T act_to_angle(const Actuator<T>& a, T motor)
{
    if (!a.linear) return motor;
    const T delta = act_abs_angle(a, Num<T>::lift(0.0f)) - act_abs_angle(a, motor);
    return a.clockwise ? a.init_angle + delta : a.init_angle - delta;
}
// ideal :111 / linear :319-370
template <typename T> __device__ __forceinline__ T act_to_motor(const Actuator<T>& a, T angle)
{
    using N = Num<T>;
    if (!a.linear) return angle;
    const T delta = a.clockwise ? angle - a.init_angle : a.init_angle - angle;
    const T init = act_abs_angle(a, N::lift(0.0f)) - delta;
    const T c = N::clamp_(N::cos_(init), -1.0f + 1e-6f, 1.0f - 1e-6f);
    T stroke = N::sqrt_(N::lift(a.offset * a.offset + a.pivot * a.pivot) - N::lift(2.0f * a.offset * a.pivot) * c);
    stroke = N::clamp_(stroke, fabsf(a.offset - a.pivot) + 1e-6f, a.offset + a.pivot - 1e-6f);
    return (stroke - a.init_stroke) * N::lift(a.increment);
}

// kinematics_rigid_body.py:194-330
template <typename T> __device__ __forceinline__ Mat4<T> kin_forward(const Params<T>& p, T motor0, T motor1)
{
    using N = Num<T>;
    const T th1 = act_to_angle(p.a1, motor0), th2 = act_to_angle(p.a2, motor1);
    Mat4<T> j1 = m_mul(m_mul(m_mul(m_rot_n(p.rot[0]), m_rot_u(p.rot[1])), m_trans(p.tr[0], p.tr[1], p.tr[2])), m_rot_e(th1));
    Mat4<T> j2 = m_mul(m_mul(m_mul(m_rot_e(p.rot[2]), m_rot_n(p.rot[3])), m_trans(p.tr[3], p.tr[4], p.tr[5])), m_rot_u(th2));
    Mat4<T> o = m_mul(m_trans(N::lift(p.pos[0]), N::lift(p.pos[1]), N::lift(p.pos[2])), j1);
    o = m_mul(o, j2);
    return m_mul(o, m_trans(p.tr[6], p.tr[7], p.tr[8]));
}

// kinematics_rigid_body.py:331-507
template <typename T> __device__ __forceinline__ void kin_inverse(const Params<T>& p, const T* normal, T& motor0, T& motor1)
{
    using N = Num<T>;
    const float eps = 1e-8f, pi = 3.14159265358979323846f;
    const Mat4<T> F1 = m_mul(m_rot_n(p.rot[0]), m_rot_u(p.rot[1]));
    const Mat4<T> F2 = m_mul(m_rot_e(p.rot[2]), m_rot_n(p.rot[3]));
    T np[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        T acc = F1.m[i] * normal[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) acc = acc + F1.m[4 * k + i] * normal[k];
        np[i] = acc;
    }
    const T f00 = F2.m[0], f01 = F2.m[1];
    const T den = N::sqrt_(f00 * f00 + f01 * f01);
    const T phi = N::atan2_(-f01, f00);
    const T ratio = N::clamp_(np[0] / (den + N::lift(eps)), -1.0f + eps, 1.0f - eps);
    T s2[2] = {N::asin_(ratio) - phi, N::lift(pi) - N::asin_(ratio) - phi}, s1[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        s2[k] = N::atan2_(N::sin_(s2[k]), N::cos_(s2[k]));
        const Mat4<T> M = m_mul(F2, m_rot_u(s2[k]));
        const T vn = -M.m[5], vu = -M.m[9];                       // M @ (0,-1,0,0)
        const T th = N::atan2_(vn * np[2] - vu * np[1], vn * np[1] + vu * np[2]);
        s1[k] = N::atan2_(N::sin_(th), N::cos_(th));
    }
    const T a0 = act_to_motor(p.a1, s1[0]), b0 = act_to_motor(p.a2, s2[0]);
    const T a1 = act_to_motor(p.a1, s1[1]), b1 = act_to_motor(p.a2, s2[1]);
    const bool ok = N::val(a0) >= p.a1.min_pos && N::val(a0) <= p.a1.max_pos && N::val(b0) >= p.a2.min_pos &&
                    N::val(b0) <= p.a2.max_pos;
    motor0 = ok ? a0 : a1;
    motor1 = ok ? b0 : b1;
}

// desired concentrator normal for the current orientation (:594-611) and the loss of :613-616
template <typename T> __device__ __forceinline__ void desired_normal(const Mat4<T>& ori, const float* incident, const float* aim,
                                                                     T* dn, T& loss)
{
    using N = Num<T>;
    T dr[3], nr = N::lift(0.0f);
#pragma unroll
    for (int k = 0; k < 3; ++k) { dr[k] = N::lift(aim[k]) - ori.m[4 * k + 3]; nr = nr + dr[k] * dr[k]; }
    nr = N::sqrt_(nr);
    if (N::val(nr) < 1e-8f) nr = N::lift(1e-8f);
    T nn = N::lift(0.0f);
#pragma unroll
    for (int k = 0; k < 3; ++k) { dn[k] = N::lift(-incident[k]) + dr[k] / nr; nn = nn + dn[k] * dn[k]; }
    nn = N::sqrt_(nn);
    if (N::val(nn) < 1e-8f) nn = N::lift(1e-8f);
    dn[3] = N::lift(0.0f);
    loss = N::lift(0.0f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < 3) dn[k] = dn[k] / nn;
        loss = loss + N::abs_(dn[k] + ori.m[4 * k + 1]);              // concentrator normal = orientation @ (0,-1,0,0)
    }
    loss = loss / N::lift(4.0f);
}

struct KinArgs {
    const float* positions;    // [H,4]
    const float* rot_dev;      // [H,4]
    const float* trans_dev;    // [H,9]
    const float* act_nonopt;   // [H,rows,2]
    const float* act_opt;      // [H,2,2] or NULL
    const float* offsets;      // [4,4]
    const float* incident;     // [H,4] (mode 1)
    const float* aim;          // [H,4] (mode 1)
    int act_rows, H, mode, max_iter;
    float min_eps;
};

// seed = index of the parameter that carries derivative 1 (-1: none): 0-3 rotation, 4-12 translation, 13-16 actuator
template <typename T> __device__ __forceinline__ Params<T> load_params(const KinArgs& a, int h, int seed)
{
    using N = Num<T>;
    Params<T> p;
    auto mk = [&](float v, int idx) { T t = N::lift(v); if constexpr (sizeof(T) == sizeof(Dual)) { if (idx == seed) reinterpret_cast<Dual&>(t).d = 1.0f; } return t; };
    for (int k = 0; k < 3; ++k) p.pos[k] = a.positions[4 * h + k];
    for (int k = 0; k < 4; ++k) p.rot[k] = mk(a.rot_dev[4 * h + k], k);
    for (int k = 0; k < 9; ++k) p.tr[k] = mk(a.trans_dev[9 * h + k], 4 + k);
    const float* no = a.act_nonopt + (int64_t)a.act_rows * 2 * h;
    Actuator<T>* acts[2] = {&p.a1, &p.a2};
    for (int c = 0; c < 2; ++c) {
        Actuator<T>& ac = *acts[c];
        ac.linear = a.act_rows >= 7;
        ac.clockwise = no[2 + c] == 1.0f;
        ac.min_pos = no[4 + c]; ac.max_pos = no[6 + c];
        if (ac.linear) {                                               // actuators_linear.py:105-178
            ac.increment = Num<float>::softplus100(no[8 + c]) + 1e-6f;
            ac.offset = Num<float>::softplus100(no[10 + c]) + 1e-6f;
            ac.pivot = Num<float>::softplus100(no[12 + c]) + 1e-6f;
            ac.init_angle = mk(a.act_opt[4 * h + c], 13 + c);
            ac.init_stroke = N::softplus100(mk(a.act_opt[4 * h + 2 + c], 15 + c)) + N::lift(1e-6f);
        } else {
            ac.increment = ac.offset = ac.pivot = 0.0f;
            ac.init_angle = N::lift(0.0f); ac.init_stroke = N::lift(0.0f);
        }
    }
    return p;
}

// One launch per forward-kinematics evaluation `it`, thread <-> heliostat.  The reference's loop is
//   evaluate -> loss -> "did ANY heliostat's loss move by more than min_eps?" -> if not: stop, else new motor positions,
// so the motor update of evaluation it-1 is done at the start of launch `it`, once the field-wide answer of launch
// it-1 (flags[it-1], written by any thread that has not converged) is visible through stream order.  A launch whose
// predecessors found the field converged returns at once: max_iter launches are always enqueued, no host round trip.
//   flags[j] (j >= 1) = 1 when some heliostat had not converged at evaluation j; evals_out = evaluations made.
__global__ __launch_bounds__(64) void rigid_body_eval_kernel(KinArgs a, int it, float* __restrict__ motor /*[H,2] in/out*/,
                                                             float* __restrict__ orientations, float* __restrict__ last_loss,
                                                             int* __restrict__ flags, int* __restrict__ evals_out)
{
    for (int j = 1; j < it; ++j)
        if (flags[j] == 0) return;                                      // converged at evaluation j: nothing more to do
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= a.H) return;
    const Params<float> p = load_params<float>(a, h, -1);
    float m0, m1;
    if (it == 0) {
        m0 = a.mode == 1 ? 0.0f : motor[2 * h];
        m1 = a.mode == 1 ? 0.0f : motor[2 * h + 1];
    } else {                                                            // next motor positions (:630-632)
        const Mat4<float> prev = kin_forward(p, motor[2 * h], motor[2 * h + 1]);
        float dn[4], loss;
        desired_normal(prev, a.incident + 4 * h, a.aim + 4 * h, dn, loss);
        kin_inverse(p, dn, m0, m1);
    }
    motor[2 * h] = m0; motor[2 * h + 1] = m1;
    if (it == a.max_iter) return;          // the reference updates the motor positions once more after its last evaluation
    const Mat4<float> ori = kin_forward(p, m0, m1);
    if (a.mode == 1) {
        float dn[4], loss;
        desired_normal(ori, a.incident + 4 * h, a.aim + 4 * h, dn, loss);
        if (it > 0 && !(fabsf(last_loss[h] - loss) <= a.min_eps)) flags[it] = 1;
        last_loss[h] = loss;
    }
    Mat4<float> off;                                                    // orientation @ initial offsets (:634)
    for (int k = 0; k < 16; ++k) off.m[k] = a.offsets[k];
    const Mat4<float> r = m_mul(ori, off);
    for (int k = 0; k < 16; ++k) orientations[16 * (int64_t)h + k] = r.m[k];
    if (h == 0) *evals_out = it + 1;
}

// thread <-> (heliostat, parameter): the chain of the forward with the recorded number of evaluations, on dual numbers
__global__ __launch_bounds__(256) void rigid_body_bwd_kernel(KinArgs a, const float* __restrict__ motor_in,
                                                             const int* __restrict__ evals_in,
                                                             const float* __restrict__ grad_orientations,
                                                             float* __restrict__ grad_rot, float* __restrict__ grad_trans,
                                                             float* __restrict__ grad_opt, float* __restrict__ grad_motor)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_par = 19;                     // 4 rotation + 9 translation + 4 actuator parameters + 2 motor positions
    if (idx >= a.H * n_par) return;
    const int h = idx / n_par, q = idx % n_par;
    const bool has_opt = a.act_rows >= 7 && grad_opt != nullptr;
    if (q >= 13 && q < 17 && !has_opt) return;
    if (q >= 17 && (a.mode != 0 || grad_motor == nullptr)) return;      // given motor positions: the calibration path only
    const Params<Dual> p = load_params<Dual>(a, h, q);
    Dual m0 = {0.0f, 0.0f}, m1 = {0.0f, 0.0f};
    Mat4<Dual> ori;
    if (a.mode == 1) {
        const int evals = *evals_in;
        for (int it = 0; it < evals; ++it) {
            ori = kin_forward(p, m0, m1);
            if (it + 1 == evals) break;
            Dual dn[4], loss;
            desired_normal(ori, a.incident + 4 * h, a.aim + 4 * h, dn, loss);
            kin_inverse(p, dn, m0, m1);
        }
    } else {
        m0 = {motor_in[2 * h], q == 17 ? 1.0f : 0.0f}; m1 = {motor_in[2 * h + 1], q == 18 ? 1.0f : 0.0f};
        ori = kin_forward(p, m0, m1);
    }
    // d(orientation @ offsets) . dL/d(orientation)
    float acc = 0.0f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float d = 0.0f;
            for (int k = 0; k < 4; ++k) d += ori.m[4 * i + k].d * a.offsets[4 * k + j];
            acc += d * grad_orientations[16 * (int64_t)h + 4 * i + j];
        }
    if (q < 4) grad_rot[4 * h + q] = acc;
    else if (q < 13) grad_trans[9 * h + (q - 4)] = acc;
    else if (q < 17) grad_opt[4 * h + (q < 15 ? q - 13 : 2 + (q - 15))] = acc;
    else grad_motor[2 * h + (q - 17)] = acc;
}

}  // namespace art

using namespace art;

static bool kin_fill(KinArgs& a, int mode, const float* positions, const float* rot_dev, const float* trans_dev,
                     const float* act_nonopt, int64_t act_rows, const float* act_opt, const float* offsets,
                     const float* incident, const float* aim, int64_t H, int max_iter, double min_eps)
{
    if (!positions || !rot_dev || !trans_dev || !act_nonopt || !offsets || H < 0 || H > (1 << 24)) return false;
    if (act_rows != 4 && act_rows != 7) return false;
    if (act_rows == 7 && !act_opt) return false;
    if (mode != 0 && mode != 1) return false;
    if (mode == 1 && (!incident || !aim || max_iter < 1)) return false;
    a.positions = positions; a.rot_dev = rot_dev; a.trans_dev = trans_dev; a.act_nonopt = act_nonopt; a.act_opt = act_opt;
    a.offsets = offsets; a.incident = incident; a.aim = aim;
    a.act_rows = (int)act_rows; a.H = (int)H; a.mode = mode; a.max_iter = mode == 1 ? max_iter : 1; a.min_eps = (float)min_eps;
    return true;
}

extern "C" int art_rigid_body_fwd(int mode, const float* positions, const float* rot_dev, const float* trans_dev,
                                  const float* act_nonopt, int64_t act_rows, const float* act_opt, const float* offsets,
                                  const float* incident, const float* aim, int64_t H, int max_iter, double min_eps,
                                  float* motor_positions, float* orientations, float* scratch, int32_t* evaluations,
                                  void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KinArgs a;
    if (H == 0) return ART_OK;                       // an empty group: nothing to read, nothing to write
    if (!motor_positions || !orientations || !scratch || !evaluations ||
        !kin_fill(a, mode, positions, rot_dev, trans_dev, act_nonopt, act_rows, act_opt, offsets, incident, aim, H, max_iter, min_eps))
        return ART_EINVAL;
    // scratch = [H] previous losses, then max_iter + 1 int flags
    float* last_loss = scratch;
    int* flags = reinterpret_cast<int*>(scratch + H);
    const int evaluations_max = mode == 1 ? max_iter : 1;
    ART_HIP(hipMemsetAsync(flags, 0, sizeof(int) * (size_t)(evaluations_max + 1), stream));
    for (int it = 0; it < evaluations_max + (mode == 1 ? 1 : 0); ++it)   // mode 1: + the trailing motor update
        hipLaunchKernelGGL(rigid_body_eval_kernel, dim3((unsigned)((H + 63) / 64)), dim3(64), 0, stream, a, it, motor_positions,
                           orientations, last_loss, flags, evaluations);
    ART_HIP(hipGetLastError());
    return ART_OK;
}

extern "C" int art_rigid_body_bwd(int mode, const float* positions, const float* rot_dev, const float* trans_dev,
                                  const float* act_nonopt, int64_t act_rows, const float* act_opt, const float* offsets,
                                  const float* incident, const float* aim, int64_t H, const float* motor_positions,
                                  const int32_t* evaluations, const float* grad_orientations, float* grad_rot_dev,
                                  float* grad_trans_dev, float* grad_act_opt, float* grad_motor_positions, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KinArgs a;
    if (H == 0) return ART_OK;
    if (!grad_orientations || !grad_rot_dev || !grad_trans_dev || !motor_positions || (mode == 1 && !evaluations) ||
        !kin_fill(a, mode, positions, rot_dev, trans_dev, act_nonopt, act_rows, act_opt, offsets, incident, aim, H, 1, 0.0))
        return ART_EINVAL;
    const int64_t n = H * 19;
    hipLaunchKernelGGL(rigid_body_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, motor_positions,
                       evaluations, grad_orientations, grad_rot_dev, grad_trans_dev, grad_act_opt, grad_motor_positions);
    ART_HIP(hipGetLastError());
    return ART_OK;
}
