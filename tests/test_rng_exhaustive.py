"""k / 2147483647.0 without a division: the three-operation form used on the device equals the
correctly rounded quotient for every possible draw, and the hemisphere coordinate with the doubling
folded into the constants equals (k / 2147483647.0) * 2.0 - 1.0.  The check is exhaustive over all 2^31 values
(a small C program, ~6 s); it runs on the CPU."""
import os
import subprocess
import textwrap

SRC = textwrap.dedent(
    r"""
    #include <math.h>
    #include <stdint.h>
    #include <stdio.h>
    int main(void) {
        const double D = 2147483647.0, c = 0x1.00000002p-31;
        const double Dh = 1073741823.5, c2 = 0x1.00000002p-30;  /* the coordinate form: doubling folded in */
        if (c != 1.0 / D) return 2;
        uint64_t bad = 0;
        for (int64_t k = 0; k <= 2147483647LL; k++) {
            double kd = (double)k, q0 = kd * c, r = fma(-q0, D, kd), q = fma(r, c, q0);
            if (q != kd / D) bad++;
            double p0 = kd * c2, rr = fma(-p0, Dh, kd), co = fma(rr, c2, p0) - 1.0;
            if (co != (kd / D) * 2.0 - 1.0) bad++;
        }
        printf("%llu\n", (unsigned long long)bad);
        return bad != 0;
    }
    """
)


def test_k_to_unit_is_exact_for_all_draws(tmp_path):
    src = tmp_path / "kdiv.c"
    exe = tmp_path / "kdiv"
    src.write_text(SRC)
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", str(src), "-o", str(exe), "-lm"], check=True, timeout=120)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "0", r.stdout
