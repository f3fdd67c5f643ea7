/* opty_hip_referee.h -- C ABI of libopty_hip_referee.so: the device side of
 * the BUILD VERIFICATION of opty_amd.
 *
 * No reference counterpart: the reference trusts its C compiler
 * (opty/utils.py:814-928 builds and imports).  hipcc 7.2 has miscompiled
 * generated kernels at the register limit, so every code object is held to
 * its expression DAG before a handle is handed out.  That machinery -- an
 * interpreter kernel and a register poisoner with 610 generated lines of
 * v_mov -- is test-and-verify infrastructure, not part of the evaluation
 * path, and lives OUTSIDE libopty_hip.so: a process that only evaluates never
 * loads it; the Python host loads it when it verifies
 * (ConstraintCollocator._verify_build).  It never supplies a value a caller
 * sees: it can only refuse a code object.
 */
#ifndef OPTY_HIP_REFEREE_H
#define OPTY_HIP_REFEREE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Build verification aid: leaves `pattern` in every vector / accumulation /
 * free scalar register of every SIMD of the current device (and waits).  A
 * referee that runs it before the kernel it checks sees a kernel that reads a
 * register it never wrote -- the hipcc 7.2 faults of DESIGN.md 4.1 -- compute
 * with the pattern instead of with the previous launch's values. */
int opty_hip_poison_registers(unsigned pattern);

/* Build verification (no reference counterpart: the reference trusts its C
 * compiler).  Evaluates an instruction tape of a problem's expression DAG
 * (opty_amd/codegen/tape.py: 8 int32 per instruction -- op, dst, a, b, c, d,
 * imm, 0) on `device`, one lane per node, over the HOST value table
 * vals[slot*nodes + node] (uploaded, run, downloaded in place; constant and
 * input slots pre-filled by the caller).  What the generated kernels of a code
 * object at the register limit are held to before a handle exists. */
int opty_hip_tape_run(int32_t device, const int32_t *code, int64_t ninstr,
                      double *vals, int64_t nslots, int64_t nodes);

/* Message of the last failing call of this library on this thread. */
const char *opty_hip_referee_last_error(void);

#ifdef __cplusplus
}
#endif

#endif /* OPTY_HIP_REFEREE_H */
