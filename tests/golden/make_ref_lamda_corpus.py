#!/usr/bin/env python3
"""A corpus of well- and ill-formed LAMDA files, and what the REFERENCE BINARY's readdata_ does with each (container only).

    python tests/golden/make_ref_lamda_corpus.py   ->  tests/golden/lamda_corpus/*.dat, tests/golden/ref_lamda_corpus.json

Row a6 of SURVEY.md section 8: readdata_ [radex.so@0x1cf90; called at emcee/pyradex/core.py:570,744; a bad file surfaces in
the reference as an exception at construction, core.py:293-298, 738-739].  The first real co.dat a user brings will not be the
two well-formed files the parser has been pinned on, so this writes 97 small mutations of tests/golden/toy6.dat (truncations,
counts that disagree with the rows, indices outside the level list, negative rates, `d` / `D` / missing exponent letters, tabs,
CRLF, values continued on the next record, a 70-level molecule, E_up <= E_low, ...) and runs the reference's own machine code on
every one of them, each in a fresh process with a fresh image (a STOP ends the process).  Recorded per file:

    outcome   "ok"                 readdata_ returned and no trapped import fired
              "stop"               the routine executed STOP (its own error path)
              "io:<what>"          libgfortran would have ended the run: end of file, a malformed number (the loader's I/O
                                   shim records these instead of raising)
              "unsupported:<what>" an input form the shim does not implement (null values, slashes, repeat counts, quoted
                                   strings): neither accepted nor rejected by this experiment
    and for "ok": nlev, nline, npart, eterm, gstat, iupp, ilow, aeinst, spfreq, xnu as parsed, and crate / ctot at
    two (T_kin, density) points -- all from the binary's COMMON blocks.

tests/test_lamda_corpus.py holds the oracle's and the product's parsers to this list: the same files accepted, bit-equal tables.
"""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

HERE = os.path.dirname(os.path.abspath(__file__))
CORPUS = os.path.join(HERE, "lamda_corpus")
TOY = open(os.path.join(HERE, "toy6.dat")).read().split("\n")
if TOY[-1] == "":
    TOY.pop()
POINTS = [(25.0, 1e4), (170.0, 3e5)]                      # (T_kin, density of every partner the file names)


def sub(lines, i, new):
    out = list(lines)
    out[i] = new
    return out


def big_molecule(nlev):
    """a ladder with nlev levels, nlev - 1 lines, one partner with all downward rates at 3 temperatures"""
    L = ["!MOLECULE", "LADDER%d" % nlev, "!MOLECULAR WEIGHT", "28.0", "!NUMBER OF ENERGY LEVELS", str(nlev), "!LEVEL + ENERGIES(cm^-1) + WEIGHT + J"]
    for j in range(nlev):
        L.append("%5d %14.6f %6.1f %5d" % (j + 1, 1.9225 * j * (j + 1), 2 * j + 1, j))
    L += ["!NUMBER OF RADIATIVE TRANSITIONS", str(nlev - 1), "!TRANS + UP + LOW + EINSTEINA(s^-1) + FREQ(GHz) + E_u(K)"]
    for j in range(1, nlev):
        L.append("%5d %5d %5d %11.3e %14.7f %10.2f" % (j, j + 1, j, 7.2e-8 * j ** 3, 115.2712 * j, 2.766 * j * (j + 1)))
    L += ["!NUMBER OF COLL PARTNERS", "1", "!COLLISIONS BETWEEN", "2 LADDER-pH2", "!NUMBER OF COLL TRANS", str(nlev * (nlev - 1) // 2),
          "!NUMBER OF COLL TEMPS", "3", "!COLL TEMPS", "  10.0  100.0  1000.0", "!TRANS + UP + LOW + COLLRATES(cm^3 s^-1)"]
    k = 0
    for u in range(2, nlev + 1):
        for lo in range(1, u):
            k += 1
            r = 3e-11 / (1 + (u - lo)) ** 1.5
            L.append("%6d %5d %5d %10.3e %10.3e %10.3e" % (k, u, lo, r, 1.3 * r, 1.9 * r))
    return L


def corpus():
    T = TOY
    lev0, lin0, col0 = 7, 16, 34                           # first level / line / rate row of toy6.dat (0-based line numbers)
    NPART, PID, NCOLL, NTEMP, TEMPS = 24, 26, 28, 30, 32   # the records holding npart, the partner id, ncoll, ntemp, the temperatures
    assert T[5] == "6" and T[14] == "7" and T[NPART] == "1" and T[PID].startswith("1 TOY6") and T[NCOLL] == "13" and T[NTEMP] == "4"
    two = (T[:24] + ["2"] + T[25:] + ["!COLLISIONS BETWEEN", "3 TOY6-oH2", "!NUMBER OF COLL TRANS", "2", "!NUMBER OF COLL TEMPS", "2",
                                        "!COLL TEMPS", " 20.0 200.0", "!TRANS + UP + LOW + COLLRATES(cm^3 s^-1)",
                                        "    1     2     1   1.0e-11  2.0e-11", "    2     6     2   3.0e-12  4.0e-12"])
    C = {
        "ok_toy6": T,
        "ok_crlf": [l + "\r" for l in T],
        "ok_tabs": [l.replace("   ", "\t") if i >= lev0 else l for i, l in enumerate(T)],
        "ok_trailing_blanks": [l + "    " for l in T],
        "ok_no_final_newline": T,                          # (written without the last newline below)
        "ok_commas": [", ".join(l.split()) if lin0 <= i < lin0 + 7 else l for i, l in enumerate(T)],
        "ok_level_rows_permuted": T[:lev0 + 1] + [T[lev0 + 2], T[lev0 + 1]] + T[lev0 + 3:],
        "ok_line_rows_permuted": T[:lin0 + 1] + [T[lin0 + 2], T[lin0 + 1]] + T[lin0 + 3:],
        "ok_rate_rows_permuted": T[:col0 + 1] + [T[col0 + 2], T[col0 + 1]] + T[col0 + 3:],
        "ok_duplicate_partner_id": (T[:24] + ["2"] + T[25:] + ["!COLLISIONS BETWEEN", "1 TOY6-H2 again", "!NUMBER OF COLL TRANS", "1", "!NUMBER OF COLL TEMPS", "2",
                                                        "!COLL TEMPS", " 20.0 200.0", "!TRANS + UP + LOW + COLLRATES(cm^3 s^-1)",
                                                        "    1     2     1   1.0e-11  2.0e-11"]),
        "bad_levels_without_qnum": [" ".join(l.split()[:3]) if lev0 <= i < lev0 + 6 else l for i, l in enumerate(T)],
        "ok_level_qnum_with_blank": sub(T, lev0 + 2, "    3      9.800000   3.0   1 1"),
        "bad_lines_without_eup": [" ".join(l.split()[:5]) if lin0 <= i < lin0 + 7 else l for i, l in enumerate(T)],
        "bad_level_number_zero": sub(T, lev0 + 2, "    0      9.800000   3.0   1_1"),
        "bad_level_number_beyond": sub(T, lev0 + 2, "    7      9.800000   3.0   1_1"),
        "bad_line_number_zero": sub(T, lin0 + 2, "    0     4     2   2.900e-05    295.2956   20.29"),
        "bad_line_number_beyond": sub(T, lin0 + 2, "    8     4     2   2.900e-05    295.2956   20.29"),
        "bad_rate_number_zero": sub(T, col0 + 2, "    0     3     2   3.5e-11  3.9e-11  4.4e-11  5.0e-11"),
        "bad_rate_number_beyond": sub(T, col0 + 2, "   14     3     2   3.5e-11  3.9e-11  4.4e-11  5.0e-11"),
        "bad_ntemp_99_declared": sub(T, NTEMP, "99"),
        "bad_ntemp_100_declared": sub(T, NTEMP, "100"),
        "bad_ncoll_99999_declared": sub(T, NCOLL, "99999"),
        "bad_ncoll_100000_declared": sub(T, NCOLL, "100000"),
        "bad_npart_nine": sub(T, NPART, "9"),
        "bad_npart_ten": sub(T, NPART, "10"),
        "bad_nline_99999_declared": sub(T, 14, "99999"),
        "bad_nline_100000_declared": sub(T, 14, "100000"),
        "bad_nlev_2999_declared": sub(T, 5, "2999"),
        "bad_level_energy_text": sub(T, lev0 + 2, "    3      abc   3.0   1_1"),
        "bad_integer_with_exponent": sub(T, lin0 + 2, "    3     4e0     2   2.900e-05    295.2956   20.29"),
        "bad_real_hex": sub(T, col0 + 2, "    3     3     2   0x1p-35  3.9e-11  4.4e-11  5.0e-11"),
        "bad_real_inf": sub(T, col0 + 2, "    3     3     2   inf  3.9e-11  4.4e-11  5.0e-11"),
        "ok_exponent_D": sub(T, col0 + 1, "    2     3     1   1.0D-11  1.4d-11  1.9E-11  2.6e-11"),
        "ok_exponent_missing_letter": sub(T, col0 + 1, "    2     3     1   1.0-11  1.4-11  1.9-11  2.6-11"),
        "ok_plus_signs": sub(T, col0 + 2, "   +3    +3    +2   +3.5e-11  3.9e-11  4.4e-11  5.0e-11"),
        "ok_extra_columns": sub(T, col0 + 3, "    4     4     1   4.0e-12  6.0e-12  9.0e-12  1.3e-11  7.7e-11 extra"),
        "ok_two_partners": two,
        "ok_partner_without_density": two,
        "ok_temps_on_two_records": T[:TEMPS] + ["   10.0   30.0", "  100.0  300.0"] + T[TEMPS + 1:],
        "ok_rate_row_on_two_records": T[:col0] + ["    1     2     1   2.1e-11  2.6e-11", "  3.3e-11  4.0e-11"] + T[col0 + 1:],
        "ok_level_row_on_two_records": T[:lev0 + 1] + ["    2      4.250000", "   3.0   1_0"] + T[lev0 + 2:],
        "ok_one_temperature": T[:NTEMP] + ["1", "!COLL TEMPS", "   30.0"] + [T[TEMPS + 1]] + [" ".join(l.split()[:4]) for l in T[col0:]],
        "ok_duplicate_rate_row": sub(T, col0 + 12, "   13     2     1   9.9e-11  9.9e-11  9.9e-11  9.9e-11"),
        "ok_upward_rate_listed": sub(T, col0 + 12, "   13     1     6   5.0e-13  9.0e-13  1.6e-12  2.8e-12"),
        "ok_temps_descending": sub(T, TEMPS, "  300.0  100.0   30.0   10.0"),
        "ok_temps_shuffled": sub(T, TEMPS, "   10.0  300.0   30.0  100.0"),
        "ok_temps_with_a_repeat": sub(T, TEMPS, "   10.0   30.0   30.0  300.0"),
        "ok_fewer_rate_rows_declared": sub(T, NCOLL, "11"),
        "ok_ladder41": big_molecule(41),
        "ok_ladder70": big_molecule(70),
        "ok_weight_line_text": sub(T, 3, "30.0   amu"),
        "ok_zero_rate": sub(T, col0 + 4, "    5     4     2   0.0  0.0  0.0  0.0"),
        "ok_leading_blank_partner_id": sub(T, PID, " 1 TOY6-H2 hand-made"),
        "ok_partner_id_2digits": sub(T, PID, "12 TOY6-H2 hand-made"),
        "ok_lower_energy_upper_level_collision": sub(T, col0 + 2, "    3     2     3   3.5e-11  3.9e-11  4.4e-11  5.0e-11"),
        "bad_empty": [],
        "bad_only_header": T[:3],
        "bad_truncated_levels": T[:lev0 + 3],
        "bad_truncated_lines": T[:lin0 + 4],
        "bad_truncated_before_partner": T[:NPART - 1],
        "bad_truncated_rates": T[:col0 + 5],
        "bad_last_rate_row_short": T[:-1] + ["   13     6     1   5.0e-13  9.0e-13"],
        "bad_more_rate_rows_declared": sub(T, NCOLL, "20"),
        "bad_nlev_zero": sub(T, 5, "0"),
        "bad_nlev_one": sub(T, 5, "1"),
        "bad_nlev_huge": sub(T, 5, "3000"),
        "bad_nlev_real": sub(T, 5, "6.0"),
        "bad_nlev_text": sub(T, 5, "six"),
        "bad_nline_zero": sub(T, 14, "0"),
        "bad_line_upper_zero": sub(T, lin0 + 2, "    3     0     2   2.900e-05    295.2956   20.29"),
        "bad_line_upper_beyond": sub(T, lin0 + 2, "    3     7     2   2.900e-05    295.2956   20.29"),
        "bad_line_lower_negative": sub(T, lin0 + 2, "    3     4    -2   2.900e-05    295.2956   20.29"),
        "bad_line_equal_energy": sub(T, lin0 + 2, "    3     2     2   2.900e-05    295.2956   20.29"),
        "bad_line_inverted": sub(T, lin0 + 2, "    3     2     4   2.900e-05    295.2956   20.29"),
        "bad_negative_einstein_a": sub(T, lin0 + 2, "    3     4     2  -2.900e-05    295.2956   20.29"),
        "bad_rate_upper_zero": sub(T, col0 + 2, "    3     0     2   3.5e-11  3.9e-11  4.4e-11  5.0e-11"),
        "bad_rate_upper_beyond": sub(T, col0 + 2, "    3     9     2   3.5e-11  3.9e-11  4.4e-11  5.0e-11"),
        "bad_rate_same_level": sub(T, col0 + 2, "    3     3     3   3.5e-11  3.9e-11  4.4e-11  5.0e-11"),
        "bad_negative_rate": sub(T, col0 + 2, "    3     3     2  -3.5e-11 -3.9e-11 -4.4e-11 -5.0e-11"),
        "bad_negative_rate_one_column": sub(T, col0 + 2, "    3     3     2   3.5e-11  3.9e-11 -4.4e-11  5.0e-11"),
        "bad_rate_text": sub(T, col0 + 2, "    3     3     2   3.5e-11  abc  4.4e-11  5.0e-11"),
        "bad_rate_nan": sub(T, col0 + 2, "    3     3     2   3.5e-11  nan  4.4e-11  5.0e-11"),
        "bad_npart_zero": sub(T, NPART, "0"),
        "bad_npart_more_than_present": sub(T, NPART, "2"),
        "bad_npart_eight": sub(T, NPART, "8"),
        "bad_partner_id_zero": sub(T, PID, "0 TOY6-?? hand-made"),
        "bad_partner_id_eight": sub(T, PID, "8 TOY6-?? hand-made"),
        "bad_partner_id_text": sub(T, PID, "H2 TOY6 hand-made"),
        "bad_ncoll_zero": sub(T, NCOLL, "0"),
        "bad_ntemp_zero": sub(T, NTEMP, "0"),
        "bad_ntemp_more_than_columns": sub(T, NTEMP, "6"),
        "bad_missing_comment_line": T[:4] + T[5:],
        "bad_binary_garbage": T[:lev0] + ["\x00\x01\x02\xff\xfe garbage \x7f"] + T[lev0 + 1:],
        "bad_very_long_line": sub(T, lev0 + 1, "    2      4.250000   3.0   " + "x" * 70000),
        "unsupported_repeat_count": sub(T, col0 + 2, "    3     3     2   4*3.5e-11"),
        "unsupported_null_value": sub(T, col0 + 2, "    3     3     2   3.5e-11,,4.4e-11  5.0e-11"),
        "unsupported_slash": sub(T, col0 + 2, "    3     3     2   3.5e-11  3.9e-11 / rest"),
    }
    return C


def densities(name, lines):
    """density by partner id: every id the file names (first character of the partner record, as (i1,a) reads it)"""
    ids = []
    for i, l in enumerate(lines):
        if l.startswith("!COLLISIONS BETWEEN") and i + 1 < len(lines):
            c = lines[i + 1][:1]
            if c.isdigit() and 1 <= int(c) <= 7:
                ids.append(int(c))
    if name == "ok_partner_without_density":
        ids = ids[:1]
    return ids or [1]


def child(path, ids, conn):
    from oracle.macho_ref import RefRadex
    R = RefRadex()
    v = R.views()
    res = {"points": []}
    real_exit = os._exit

    def exit_with_report(code):                            # (the loader's STOP trap ends the process: say what was written first)
        conn.send({"stopped": True, "messages": list(R.io.messages) + ["STOP " + getattr(R, "stop_message", "")],
                   "trap_log": [t for t in R.trap_log if t != "_gfortran_stop_string"]})
        real_exit(code)
    os._exit = exit_with_report
    for k, (tkin, dens) in enumerate(POINTS):
        R.readdata(path, tkin, {i: dens for i in ids})
        if R.trap_log:
            break
        n, L = int(v["imolec_hdr"][0]), int(v["imolec_hdr"][1])
        if k == 0:
            res.update(nlev=n, nline=L, npart=int(v["imolec_hdr"][3]),
                       eterm=[float(x) for x in v["eterm"][:n]], gstat=[float(x) for x in v["gstat"][:n]],
                       iupp=[int(x) for x in v["iupp"][:L]], ilow=[int(x) for x in v["ilow"][:L]],
                       aeinst=[float(x) for x in v["aeinst"][:L]], spfreq=[float(x) for x in v["spfreq"][:L]],
                       xnu=[float(x) for x in v["xnu"][:L]])
        from oracle.macho_ref import MAXLEV
        cr = v["crate"].reshape(MAXLEV, MAXLEV).T[:n, :n]            # crate(i,j), column-major
        res["points"].append(dict(tkin=tkin, density={str(i): dens for i in ids}, crate=[float(x) for x in cr.ravel()],
                                  ctot=[float(x) for x in v["ctot"][:n]]))
    res["trap_log"] = list(R.trap_log)
    res["messages"] = list(R.io.messages)
    os._exit = real_exit
    conn.send(res)
    conn.close()


def classify(trap_log):
    if not trap_log:
        return "ok"
    t = trap_log[0]
    if "not implemented" in t or "null value" in t:
        return "unsupported:" + t.replace("fortran-io: ", "")
    if t.startswith("fortran-io:"):
        return "io:" + t.replace("fortran-io: ", "")
    return "trap:" + t


def main():
    os.makedirs(CORPUS, exist_ok=True)
    ctx = mp.get_context("fork")
    out = {}
    for name, lines in corpus().items():
        path = os.path.join(CORPUS, name + ".dat")
        text = "\n".join(lines) + ("" if name == "ok_no_final_newline" or not lines else "\n")
        with open(path, "w", encoding="latin-1", newline="") as f:
            f.write(text)
        ids = densities(name, lines)
        a, b = ctx.Pipe(False)
        p = ctx.Process(target=child, args=(os.path.relpath(path, ROOT), ids, b))
        cwd = os.getcwd()
        os.chdir(ROOT)                                       # (impex.molfile holds 120 characters: relative paths)
        p.start()
        os.chdir(cwd)
        b.close()
        try:
            res = a.recv() if a.poll(120) else None
        except EOFError:
            res = None
        p.join(10)
        if res is None or res.get("stopped"):
            # (an I/O condition the shim recorded BEFORE the routine stopped is the outcome: real libgfortran ends the run there)
            rec = {"outcome": classify(res["trap_log"]) if res and res["trap_log"] else ("stop" if p.exitcode == 97 else "died:%s" % p.exitcode)}
            if res and res["messages"]:
                rec["messages"] = res["messages"][-4:]
        else:
            rec = {"outcome": classify(res["trap_log"])}
            if rec["outcome"] == "ok":
                rec.update({k: res[k] for k in ("nlev", "nline", "npart", "eterm", "gstat", "iupp", "ilow", "aeinst", "spfreq", "xnu", "points")})
            if res["messages"]:
                rec["messages"] = res["messages"][:4]
        rec["density_ids"] = ids
        out[name] = rec
        print("%-42s %-40s %s" % (name, rec["outcome"], "; ".join(rec.get("messages", []))[:100]), flush=True)
    json.dump(dict(source="radex.so:_readdata_ on tests/golden/lamda_corpus/*.dat (make_ref_lamda_corpus.py)",
                   points=POINTS, files=out), open(os.path.join(HERE, "ref_lamda_corpus.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
