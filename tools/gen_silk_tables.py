#!/usr/bin/env python3
"""Write concentus_amd/csrc/silk_nlsf_tables.h: the two NLSF codebooks of the Opus specification
(RFC 6716 section 4.2.7.5; the reference holds them in opus-fix/silk/tables_NLSF_CB_WB.c and
tables_NLSF_CB_NB_MB.c as `silk_NLSF_CB_WB` / `silk_NLSF_CB_NB_MB`, struct silk_NLSF_CB_struct of
silk/structs.h:83-95).

These are trained constants of the bitstream format and cannot be derived: the numbers are read out of the
compiled, unmodified reference (oracle/_ref/libopus_ref.so) through its exported symbols and written as
literal arrays; tests/test_tables.py re-reads the library and compares when it is present.

Usage: python tools/gen_silk_tables.py   (needs `make -C oracle ref`)
"""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFLIB = os.path.join(ROOT, "oracle", "_ref", "libopus_ref.so")
OUT = os.path.join(ROOT, "concentus_amd", "csrc", "silk_nlsf_tables.h")
NLSF_QUANT_MAX_AMPLITUDE = 4            # silk/define.h
EC_TABLES = 8                           # entropy-table selectors are 3 bits (NLSF_unpack.c:48-51)


class CBStruct(C.Structure):            # silk/structs.h:83-95 (x86-64 layout)
    _fields_ = [("nVectors", C.c_int16), ("order", C.c_int16), ("quantStepSize_Q16", C.c_int16),
                ("invQuantStepSize_Q6", C.c_int16),
                ("CB1_NLSF_Q8", C.POINTER(C.c_uint8)), ("CB1_iCDF", C.POINTER(C.c_uint8)),
                ("pred_Q8", C.POINTER(C.c_uint8)), ("ec_sel", C.POINTER(C.c_uint8)),
                ("ec_iCDF", C.POINTER(C.c_uint8)), ("ec_Rates_Q5", C.POINTER(C.c_uint8)),
                ("deltaMin_Q15", C.POINTER(C.c_int16))]


def read_codebook(lib, symbol):
    cb = CBStruct.in_dll(lib, symbol)
    nv, order = cb.nVectors, cb.order
    nec = EC_TABLES * (2 * NLSF_QUANT_MAX_AMPLITUDE + 1)
    return {
        "nVectors": nv, "order": order, "quantStepSize_Q16": cb.quantStepSize_Q16,
        "invQuantStepSize_Q6": cb.invQuantStepSize_Q6,
        "CB1_NLSF_Q8": [cb.CB1_NLSF_Q8[i] for i in range(nv * order)],
        "CB1_iCDF": [cb.CB1_iCDF[i] for i in range(2 * nv)],
        "pred_Q8": [cb.pred_Q8[i] for i in range(2 * (order - 1))],
        "ec_sel": [cb.ec_sel[i] for i in range(nv * order // 2)],
        "ec_iCDF": [cb.ec_iCDF[i] for i in range(nec)],
        "ec_Rates_Q5": [cb.ec_Rates_Q5[i] for i in range(nec)],
        "deltaMin_Q15": [cb.deltaMin_Q15[i] for i in range(order + 1)],
    }


def codebooks():
    lib = C.CDLL(REFLIB)
    return {"WB": read_codebook(lib, "silk_NLSF_CB_WB"), "NB_MB": read_codebook(lib, "silk_NLSF_CB_NB_MB")}


def ltp_tables():
    """The three LTP gain codebooks (RFC 6716 section 4.2.7.6.2; opus-fix/silk/tables_LTP.c): vectors, code lengths, gains."""
    lib = C.CDLL(REFLIB)
    sizes = list((C.c_int8 * 3).in_dll(lib, "silk_LTP_vq_sizes"))
    vq = (C.POINTER(C.c_int8) * 3).in_dll(lib, "silk_LTP_vq_ptrs_Q7")
    bits = (C.POINTER(C.c_uint8) * 3).in_dll(lib, "silk_LTP_gain_BITS_Q5_ptrs")
    gain = (C.POINTER(C.c_uint8) * 3).in_dll(lib, "silk_LTP_vq_gain_ptrs_Q7")
    out = {"sizes": sizes, "middle_avg_RD_Q14": C.c_int16.in_dll(lib, "silk_LTP_gain_middle_avg_RD_Q14").value,
           "LTPScales_table_Q14": list((C.c_int16 * 3).in_dll(lib, "silk_LTPScales_table_Q14"))}
    for k in range(3):
        out["vq%d" % k] = [vq[k][i] for i in range(sizes[k] * 5)]
        out["bits%d" % k] = [bits[k][i] for i in range(sizes[k])]
        out["gain%d" % k] = [gain[k][i] for i in range(sizes[k])]
    return out


ENTROPY_TABLES = (   # (symbol in the reference, entries): the iCDF / code-length tables silk_encode_indices and silk_encode_pulses use
    ("silk_gain_iCDF", 24), ("silk_delta_gain_iCDF", 41), ("silk_pitch_lag_iCDF", 32), ("silk_pitch_delta_iCDF", 21),
    ("silk_pitch_contour_iCDF", 34), ("silk_pitch_contour_NB_iCDF", 11), ("silk_pitch_contour_10_ms_iCDF", 12),
    ("silk_pitch_contour_10_ms_NB_iCDF", 3), ("silk_type_offset_VAD_iCDF", 4), ("silk_type_offset_no_VAD_iCDF", 2),
    ("silk_NLSF_interpolation_factor_iCDF", 5), ("silk_NLSF_EXT_iCDF", 7), ("silk_uniform3_iCDF", 3), ("silk_uniform4_iCDF", 4),
    ("silk_uniform5_iCDF", 5), ("silk_uniform6_iCDF", 6), ("silk_uniform8_iCDF", 8), ("silk_LTP_per_index_iCDF", 3),
    ("silk_LTPscale_iCDF", 3), ("silk_lsb_iCDF", 2), ("silk_max_pulses_table", 4), ("silk_pulses_per_block_iCDF", 180),
    ("silk_pulses_per_block_BITS_Q5", 162), ("silk_rate_levels_iCDF", 18), ("silk_rate_levels_BITS_Q5", 18),
    ("silk_shell_code_table0", 152), ("silk_shell_code_table1", 152), ("silk_shell_code_table2", 152), ("silk_shell_code_table3", 152),
    ("silk_shell_code_table_offsets", 17), ("silk_sign_iCDF", 42),
)


def entropy_tables():
    """name -> list, plus the three LTP gain iCDFs (behind silk_LTP_gain_iCDF_ptrs) flattened as 'silk_LTP_gain_iCDF' (8 + 16 + 32)."""
    lib = C.CDLL(REFLIB)
    out = {name: list((C.c_uint8 * n).in_dll(lib, name)) for name, n in ENTROPY_TABLES}
    ptrs = (C.POINTER(C.c_uint8) * 3).in_dll(lib, "silk_LTP_gain_iCDF_ptrs")
    out["silk_LTP_gain_iCDF"] = [ptrs[k][i] for k in range(3) for i in range(8 << k)]
    return out


def emit_entropy(path):
    t = entropy_tables()
    o = ["// silk_entropy_tables.h -- GENERATED by tools/gen_silk_tables.py; do not edit.",
         "// The probability models (iCDFs) and code-length tables of the SILK side information and excitation coding, constants of the",
         "// Opus specification (RFC 6716 sections 4.2.7.3 - 4.2.7.8; opus-fix/silk/tables_gain.c, tables_pitch_lag.c, tables_other.c,",
         "// tables_LTP.c, tables_pulses_per_block.c), read out of the compiled reference; tests/test_tables.py compares every array.",
         "#pragma once", '#include "wave.h"', "", "namespace ca {", ""]
    for name, vals in t.items():
        o.append(_arr("u8", "SILK_" + name[5:], vals, 18 if "per_block" in name else 19 if "shell_code_table" in name and "offsets" not in name else 16))
    o += ["", "}  // namespace ca", ""]
    with open(path, "w") as f:
        f.write("\n".join(o))
    print("wrote", path)


def _arr(ctype, name, vals, per_line):
    lines = ["CA_DEVICE_CONST %s %s[%d] = {" % (ctype, name, len(vals))]
    for i in range(0, len(vals), per_line):
        lines.append("    " + ", ".join("%d" % v for v in vals[i:i + per_line]) + ",")
    lines.append("};")
    return "\n".join(lines)


def main():
    cbs = codebooks()
    o = ["// silk_nlsf_tables.h -- GENERATED by tools/gen_silk_tables.py; do not edit.",
         "// The NLSF codebooks of the Opus specification (RFC 6716 section 4.2.7.5; the reference's silk_NLSF_CB_WB and",
         "// silk_NLSF_CB_NB_MB, opus-fix/silk/tables_NLSF_CB_WB.c / tables_NLSF_CB_NB_MB.c), flattened to plain arrays.",
         "// tests/test_tables.py compares every array with the compiled reference.",
         "#pragma once", '#include "wave.h"', "", "namespace ca {", ""]
    for tag, cb in cbs.items():
        p = "SILK_NLSF_%s_" % tag
        o.append("// ---- %s: %d vectors of order %d" % (tag, cb["nVectors"], cb["order"]))
        o.append("enum { %sNVECTORS = %d, %sORDER = %d, %sQUANT_STEP_Q16 = %d, %sINV_QUANT_STEP_Q6 = %d };"
                 % (p, cb["nVectors"], p, cb["order"], p, cb["quantStepSize_Q16"], p, cb["invQuantStepSize_Q6"]))
        o.append(_arr("u8", p + "CB1_Q8", cb["CB1_NLSF_Q8"], cb["order"]))
        o.append(_arr("u8", p + "CB1_iCDF", cb["CB1_iCDF"], 16))
        o.append(_arr("u8", p + "pred_Q8", cb["pred_Q8"], cb["order"] - 1))
        o.append(_arr("u8", p + "ec_sel", cb["ec_sel"], cb["order"] // 2))
        o.append(_arr("u8", p + "ec_iCDF", cb["ec_iCDF"], 9))
        o.append(_arr("u8", p + "ec_Rates_Q5", cb["ec_Rates_Q5"], 9))
        o.append(_arr("i16", p + "deltaMin_Q15", cb["deltaMin_Q15"], 17))
        o.append("")
    lt = ltp_tables()
    o.append("// ---- LTP gain codebooks (RFC 6716 section 4.2.7.6.2; opus-fix/silk/tables_LTP.c) and the LTP state scaling table")
    o.append("// (opus-fix/silk/tables_other.c:100). Codebook k starts at SILK_LTP_CB_OFF[k] in the flat arrays.")
    off = [0, lt["sizes"][0], lt["sizes"][0] + lt["sizes"][1]]
    o.append("enum { SILK_LTP_GAIN_MIDDLE_AVG_RD_Q14 = %d };" % lt["middle_avg_RD_Q14"])
    o.append(_arr("u8", "SILK_LTP_CB_SIZE", lt["sizes"], 3))
    o.append(_arr("u8", "SILK_LTP_CB_OFF", off, 3))
    o.append(_arr("i8", "SILK_LTP_VQ_Q7", lt["vq0"] + lt["vq1"] + lt["vq2"], 5))
    o.append(_arr("u8", "SILK_LTP_BITS_Q5", lt["bits0"] + lt["bits1"] + lt["bits2"], 16))
    o.append(_arr("u8", "SILK_LTP_VQ_GAIN_Q7", lt["gain0"] + lt["gain1"] + lt["gain2"], 16))
    o.append(_arr("i16", "SILK_LTPScales_table_Q14", lt["LTPScales_table_Q14"], 3))
    o.append("")
    o += ["}  // namespace ca", ""]
    with open(OUT, "w") as f:
        f.write("\n".join(o))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
    emit_entropy(os.path.join(os.path.dirname(OUT), "silk_entropy_tables.h"))
