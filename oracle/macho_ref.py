"""Run the reference's own RADEX machine code (x86-64 Mach-O bundle) on Linux.

TEST INFRASTRUCTURE ONLY, container-only: it reads
/root/reference/emcee/pyradex/radex/radex.so, which does not exist on the GPU
box.  It is used by tests/golden/make_ref_vectors.py to produce golden
input/output vectors (committed as JSON under tests/golden/) that pin the CPU
oracle against the reference binary itself.

How: the reference ships RADEX only as a macOS Mach-O bundle.  macOS x86-64 and
Linux x86-64 share the System V calling convention and the code is
position-independent, so the numerical routines (matrix_, escprob_, backrad_,
lubksb_ -> sgeir_/sgefa_/sgesl_) can be executed in-process once the image is
mapped:  segments are mmap'ed at one contiguous slide, rebase fix-ups applied,
and the lazy/non-lazy symbol pointers bound.  libm/libc imports (exp, log,
log10, pow, malloc, free, memcpy, memset, __bzero) are bound to glibc.  Every
other import (CPython API, STOP, ...) is bound to a *trap* that records
its name; a vector is only accepted when no trap fired, i.e. when no error
path was executed.

readdata_ (called by the reference at emcee/pyradex/core.py:570,744) reads the
LAMDA file through libgfortran.3's OPEN / list-directed READ / CLOSE.  Those
eleven entry points (st_open, st_read, transfer_integer/real/character,
st_read_done, st_close and the WRITE family used for its warnings) are served
by `_FortranIO` below: a minimal implementation of exactly the statement forms
readdata_ compiles to (OPEN(unit, file=, status='old'), READ(unit,*) items,
READ(unit,'(a)') string, READ(unit,'(i1,a)') id, string), operating on the st_parameter_* fields the binary
itself fills in (offsets read from the disassembly at radex.so@0x1cfaf-0x1d014,
@0x1d065-0x1d0ab).  Decimal -> double conversion is Python's float(), which is
correctly rounded like the strtod libgfortran calls.  Anything outside those
forms (repeat counts, null values, slashes, end of file, a malformed number)
is recorded as a trap, so such a vector is rejected instead of guessed.
The numbers readdata_ produces from the file -- xnu, crate, ctot -- are then
the reference's own (tests/golden/make_ref_readdata.py -> ref_readdata.json).
Other COMMON state is populated directly, exactly as pyradex itself pokes it
(emcee/pyradex/core.py:476-482, 853-854).

Nothing from the reference is copied into the repo: only numbers it computes.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import mmap
import os
import struct

import numpy as np

REF_SO = "/root/reference/emcee/pyradex/radex/radex.so"

LC_SEGMENT_64 = 0x19
LC_SYMTAB = 0x2
LC_DYSYMTAB = 0xB
LC_DYLD_INFO = 0x22
LC_DYLD_INFO_ONLY = 0x80000022

S_NON_LAZY_SYMBOL_POINTERS = 0x6
S_LAZY_SYMBOL_POINTERS = 0x7
INDIRECT_SYMBOL_LOCAL = 0x80000000
INDIRECT_SYMBOL_ABS = 0x40000000

MAXLEV = 2999
MAXLINE = 99999


def available() -> bool:
    return os.path.exists(REF_SO) and os.uname().machine == "x86_64"


class MachO:
    def __init__(self, path: str = REF_SO):
        with open(path, "rb") as f:
            self.data = f.read()
        d = self.data
        magic, cputype, _, filetype, ncmds, _, _, _ = struct.unpack_from("<IiiIIIII", d, 0)
        if magic != 0xFEEDFACF or cputype != 0x01000007:
            raise ValueError("not an x86-64 Mach-O image")
        off = 32
        self.segments = []
        self.sections = []
        self.symtab = None
        self.dysymtab = None
        self.dyld_info = None
        for _ in range(ncmds):
            cmd, cmdsize = struct.unpack_from("<II", d, off)
            if cmd == LC_SEGMENT_64:
                segname = d[off + 8:off + 24].rstrip(b"\0").decode()
                vmaddr, vmsize, fileoff, filesize, _, _, nsects, _ = struct.unpack_from(
                    "<QQQQiiII", d, off + 24)
                self.segments.append((segname, vmaddr, vmsize, fileoff, filesize))
                so = off + 72
                for _s in range(nsects):
                    sectname = d[so:so + 16].rstrip(b"\0").decode()
                    addr, size, _o, _a, _r, _n, flags, res1, res2, _ = struct.unpack_from(
                        "<QQIIIIIIII", d, so + 32)
                    self.sections.append((sectname, addr, size, flags, res1, res2))
                    so += 80
            elif cmd == LC_SYMTAB:
                self.symtab = struct.unpack_from("<IIII", d, off + 8)
            elif cmd == LC_DYSYMTAB:
                self.dysymtab = struct.unpack_from("<" + "I" * 18, d, off + 8)
            elif cmd in (LC_DYLD_INFO, LC_DYLD_INFO_ONLY):
                self.dyld_info = struct.unpack_from("<" + "I" * 10, d, off + 8)
            off += cmdsize
        symoff, nsyms, stroff, _strsize = self.symtab
        self.symbols = []
        self.defined = {}
        for i in range(nsyms):
            n_strx, n_type, n_sect, _n_desc, n_value = struct.unpack_from("<IBBHQ", d, symoff + 16 * i)
            end = d.index(b"\0", stroff + n_strx)
            name = d[stroff + n_strx:end].decode()
            self.symbols.append((name, n_type, n_sect, n_value))
            if (n_type & 0x0E) == 0x0E and not (n_type & 0xE0):   # N_SECT, not a stab
                self.defined[name] = n_value
        indoff, nind = self.dysymtab[12], self.dysymtab[13]
        self.indirect = struct.unpack_from("<%dI" % nind, d, indoff)
        self.vmsize = max(v + s for _, v, s, _, _ in self.segments)


class _FortranIO:
    """libgfortran.3 I/O entry points as readdata_ uses them (see the module docstring).

    st_parameter_common: flags u32 @0, unit i32 @4, filename @8, line i32 @16.
    st_parameter_open:   file_len i32 @0x2c, file @0x30, status @0x38, status_len i32 @0x40.
    st_parameter_dt:     format @0x48, format_len i32 @0x50.
    flags: 0x80 = list-directed, 0x1000 = has a format; low two bits = library return code."""

    F_LIST, F_FORMAT = 0x80, 0x1000

    def __init__(self, trap_log):
        self.trap_log = trap_log
        self.units = {}            # unit -> [records, next record]
        self.stmt = {}             # address of the statement's parameter block -> state
        self.messages = []         # what the routine WROTE (its warnings), for information
        self.opened = []

    @staticmethod
    def _i32(addr):
        return C.c_int32.from_address(addr)

    @staticmethod
    def _u32(addr):
        return C.c_uint32.from_address(addr)

    def _bad(self, what):
        self.trap_log.append("fortran-io: " + what)

    # ---- OPEN / CLOSE -----------------------------------------------------------------------------
    def st_open(self, dt):
        n = self._i32(dt + 0x2c).value
        name = C.string_at(C.c_void_p.from_address(dt + 0x30).value, n).decode("latin-1").rstrip(" ")
        unit = self._i32(dt + 4).value
        flags = self._u32(dt)
        try:
            with open(name, "r", newline="", encoding="latin-1") as f:      # (bytes, as a Fortran unit sees them)
                recs = f.read().split("\n")
        except OSError:
            flags.value = (flags.value & ~3) | 1                      # IOPARM_LIBRETURN_ERROR: the routine's own err= path
            return
        if recs and recs[-1] == "":
            recs.pop()
        self.units[unit] = [[r.rstrip("\r") for r in recs], 0]
        self.opened.append(name)
        flags.value &= ~3

    def st_close(self, dt):
        self.units.pop(self._i32(dt + 4).value, None)

    # ---- READ -----------------------------------------------------------------------------------------
    def st_read(self, dt):
        unit, flags = self._i32(dt + 4).value, self._u32(dt).value
        if unit not in self.units:
            self._bad("READ on unit %d which is not open" % unit)
        if not flags & (self.F_LIST | self.F_FORMAT) or flags & ~(self.F_LIST | self.F_FORMAT | 3):
            self._bad("READ statement form %#x not implemented" % flags)
        fmt = None
        if flags & self.F_FORMAT:
            fmt = C.string_at(C.c_void_p.from_address(dt + 0x48).value, self._i32(dt + 0x50).value).decode().strip().lower()
            if fmt not in ("(a)", "(i1,a)"):
                self._bad("format %r not implemented" % fmt)
        self.stmt[dt] = dict(unit=unit, fmt=fmt, rec=None, col=0)

    def _next_record(self, st):
        u = self.units.get(st["unit"])
        if u is None or u[1] >= len(u[0]):
            self._bad("end of file")
            st["rec"], st["col"] = "", 0
            return
        st["rec"], st["col"] = u[0][u[1]], 0
        u[1] += 1

    def _token(self, st):
        """next list-directed item: blanks separate, ONE comma may follow an item; records are consumed as needed"""
        for _ in range(100000):
            if st["rec"] is None:
                self._next_record(st)
            rec, c = st["rec"], st["col"]
            while c < len(rec) and rec[c] in " \t":
                c += 1
            if c >= len(rec):
                if "fortran-io: end of file" in self.trap_log:
                    return "0"
                st["rec"] = None
                continue
            if rec[c] in ",/":
                self._bad("null value or slash in list-directed input")
                return "0"
            if rec[c] in "'\"":
                q = rec[c]
                e = rec.find(q, c + 1)
                if e < 0 or (e + 1 < len(rec) and rec[e + 1] == q):
                    self._bad("delimited string form not implemented")
                    return ""
                tok, c = rec[c + 1:e], e + 1
            else:
                b = c
                while c < len(rec) and rec[c] not in " \t,/":
                    c += 1
                tok = rec[b:c]
            while c < len(rec) and rec[c] in " \t":
                c += 1
            if c < len(rec) and rec[c] == ",":
                c += 1
            st["col"] = c
            return tok
        self._bad("runaway read")
        return "0"

    def transfer_integer(self, dt, p, kind):
        st = self.stmt[dt]
        if st["fmt"] == "(i1,a)" and st["rec"] is None:               # I1: the first column of a fresh record
            self._next_record(st)
            tok, st["col"] = st["rec"][:1].replace(" ", "0") or "0", 1   # (blanks read as zero: BN is not in effect... nor needed)
        elif st["fmt"] is not None:
            self._bad("integer item under format %r not implemented" % st["fmt"])
            return
        else:
            tok = self._token(st)
        t = tok[1:] if tok[:1] in "+-" else tok
        if kind != 4 or not t.isdigit():
            self._bad("bad integer %r" % tok)
            return
        C.c_int32.from_address(p).value = int(tok)

    def transfer_real(self, dt, p, kind):
        tok = self._token(self.stmt[dt])
        t = tok.lower().replace("d", "e")
        # (libgfortran's read_real also takes an exponent introduced by its sign alone: 1.0-11 = 1.0e-11)
        k = max(t.rfind("+"), t.rfind("-"))
        if k > 0 and "e" not in t and t[k - 1] in "0123456789.":
            t = t[:k] + "e" + t[k:]
        try:
            if kind != 8 or "*" in t or not t or t.strip("+-.0123456789e") or "inf" in t or "nan" in t:
                raise ValueError(t)
            v = float(t)                                               # correctly rounded, like strtod
        except ValueError:
            self._bad("bad real %r" % tok)
            return
        C.c_double.from_address(p).value = v

    def transfer_character(self, dt, p, n):
        st = self.stmt[dt]
        if st["fmt"] == "(a)":
            self._next_record(st)
            txt, st["col"] = st["rec"], len(st["rec"])
        elif st["fmt"] == "(i1,a)":                                    # A without a width: the variable's length
            if st["rec"] is None or st["col"] != 1:
                self._bad("character item of (i1,a) out of order")
                return
            txt, st["col"] = st["rec"][1:1 + n], 1 + n
        else:
            txt = self._token(st)
        C.memmove(p, txt[:n].ljust(n).encode("latin-1"), n)

    def st_read_done(self, dt):
        st = self.stmt.pop(dt, None)
        if st is not None and st["rec"] is None and not any("end of file" in t for t in self.trap_log):
            self._next_record(st)                                      # READ without items: skips one record

    # ---- WRITE (warnings such as "Tkin lower than..."): recorded, never an error -----------------------
    def st_write(self, dt):
        self.stmt[dt] = dict(out=[])

    def transfer_character_write(self, dt, p, n):
        self.stmt.setdefault(dt, dict(out=[]))["out"].append(C.string_at(p, n).decode("latin-1"))

    def transfer_integer_write(self, dt, p, kind):
        v = C.c_int32.from_address(p).value if kind == 4 else C.c_int64.from_address(p).value
        self.stmt.setdefault(dt, dict(out=[]))["out"].append(str(v))

    def transfer_real_write(self, dt, p, kind):
        v = C.c_double.from_address(p).value if kind == 8 else C.c_float.from_address(p).value
        self.stmt.setdefault(dt, dict(out=[]))["out"].append(repr(v))

    def st_write_done(self, dt):
        st = self.stmt.pop(dt, None)
        if st:
            self.messages.append(" ".join(st["out"]))


def _uleb(d, p):
    r = 0
    sh = 0
    while True:
        b = d[p]
        p += 1
        r |= (b & 0x7F) << sh
        sh += 7
        if not b & 0x80:
            return r, p


class RefRadex:
    """The reference RADEX image mapped into this process."""

    def __init__(self, path: str = REF_SO):
        self.m = MachO(path)
        libc = C.CDLL(None, use_errno=True)
        libc.mmap.restype = C.c_void_p
        libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
        size = (self.m.vmsize + 0xFFFF) & ~0xFFFF
        MAP_NORESERVE = 0x4000
        base = libc.mmap(None, size, mmap.PROT_READ | mmap.PROT_WRITE | mmap.PROT_EXEC,
                         mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS | MAP_NORESERVE, -1, 0)
        if base in (None, C.c_void_p(-1).value):
            raise OSError(C.get_errno(), "mmap of %d bytes failed" % size)
        self.base = base
        self.size = size
        for name, vmaddr, _vmsize, fileoff, filesize in self.m.segments:
            if name == "__LINKEDIT" or filesize == 0:
                continue
            C.memmove(base + vmaddr, self.m.data[fileoff:fileoff + filesize], filesize)
        self._rebase()
        self.trap_log = []
        self._keep = []
        self.io = _FortranIO(self.trap_log)
        self._bind(libc)
        self._prototypes()

    # --- dyld emulation --------------------------------------------------
    def _rebase(self):
        d = self.m.data
        off, size = self.m.dyld_info[0], self.m.dyld_info[1]
        p, end = off, off + size
        seg_addr = 0
        addr = 0
        segs = self.m.segments

        def fix(a):
            ptr = C.c_uint64.from_address(self.base + a)
            ptr.value = ptr.value + self.base

        while p < end:
            byte = d[p]
            p += 1
            op, imm = byte & 0xF0, byte & 0x0F
            if op == 0x00:        # DONE
                break
            elif op == 0x10:      # SET_TYPE_IMM
                pass
            elif op == 0x20:      # SET_SEGMENT_AND_OFFSET_ULEB
                v, p = _uleb(d, p)
                seg_addr = segs[imm][1]
                addr = seg_addr + v
            elif op == 0x30:      # ADD_ADDR_ULEB
                v, p = _uleb(d, p)
                addr += v
            elif op == 0x40:      # ADD_ADDR_IMM_SCALED
                addr += imm * 8
            elif op == 0x50:      # DO_REBASE_IMM_TIMES
                for _ in range(imm):
                    fix(addr)
                    addr += 8
            elif op == 0x60:      # DO_REBASE_ULEB_TIMES
                n, p = _uleb(d, p)
                for _ in range(n):
                    fix(addr)
                    addr += 8
            elif op == 0x70:      # DO_REBASE_ADD_ADDR_ULEB
                v, p = _uleb(d, p)
                fix(addr)
                addr += 8 + v
            elif op == 0x80:      # DO_REBASE_ULEB_TIMES_SKIPPING_ULEB
                n, p = _uleb(d, p)
                sk, p = _uleb(d, p)
                for _ in range(n):
                    fix(addr)
                    addr += 8 + sk
            else:
                raise ValueError("bad rebase opcode %#x" % byte)

    def _make_trap(self, name):
        log = self.trap_log

        def trap(*_a):
            log.append(name)
            if name in ("_gfortran_stop_string", "__stack_chk_fail"):
                os.write(2, ("macho_ref: fatal import %s called\n" % name).encode())
                os._exit(97)
            return 0
        if name == "_gfortran_stop_string":                      # STOP 'text': (const char *, int) -- keep the text
            def stop(msg, n):
                self.stop_message = C.string_at(msg, n).decode("latin-1") if msg and n > 0 else ""
                return trap()
            cb = C.CFUNCTYPE(C.c_long, C.c_void_p, C.c_int)(stop)
            self._keep.append(cb)
            return C.cast(cb, C.c_void_p).value
        cb = C.CFUNCTYPE(C.c_long)(trap)
        self._keep.append(cb)
        return C.cast(cb, C.c_void_p).value

    def _make_io(self, name):
        """one libgfortran entry point served by _FortranIO: (dt) or (dt, pointer, kind / length)"""
        fn = getattr(self.io, name)
        if name.startswith("transfer_"):
            cb = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int)(lambda dt, p, k: fn(dt, p, k))
        else:
            cb = C.CFUNCTYPE(None, C.c_void_p)(lambda dt: fn(dt))
        self._keep.append(cb)
        return C.cast(cb, C.c_void_p).value

    def _bind(self, libc):
        libm = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        real = {"exp": libm, "log": libm, "log10": libm, "pow": libm,
                "malloc": libc, "free": libc, "memcpy": libc, "memset": libc,
                "memcmp": libc, "strlen": libc, "strcmp": libc, "strncpy": libc}
        guard = (C.c_uint64 * 2)(0x5A5A5A5A00C0FFEE, 0)
        self._keep.append(guard)
        syms = self.m.symbols
        for sectname, addr, size, flags, res1, _res2 in self.m.sections:
            stype = flags & 0xFF
            if stype not in (S_LAZY_SYMBOL_POINTERS, S_NON_LAZY_SYMBOL_POINTERS):
                continue
            for i in range(size // 8):
                isym = self.m.indirect[res1 + i]
                slot = C.c_uint64.from_address(self.base + addr + 8 * i)
                if isym & (INDIRECT_SYMBOL_LOCAL | INDIRECT_SYMBOL_ABS):
                    continue          # rebased already / absolute
                name = syms[isym][0]
                bare = name[1:] if name.startswith("_") else name
                if name in self.m.defined:
                    slot.value = self.base + self.m.defined[name]
                elif bare in real:
                    slot.value = C.cast(getattr(real[bare], bare), C.c_void_p).value
                elif bare == "__bzero":
                    slot.value = C.cast(libc.bzero, C.c_void_p).value
                elif bare == "__stack_chk_guard":
                    slot.value = C.addressof(guard)
                elif bare.startswith("_gfortran_") and hasattr(self.io, bare[len("_gfortran_"):]):
                    slot.value = self._make_io(bare[len("_gfortran_"):])
                else:
                    slot.value = self._make_trap(bare)

    # --- symbol access ------------------------------------------------------
    def addr(self, name: str) -> int:
        return self.base + self.m.defined[name]

    def _fn(self, name, restype, *argtypes):
        return C.CFUNCTYPE(restype, *argtypes)(self.addr(name))

    def _prototypes(self):
        pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int)
        self.f_matrix = self._fn("_matrix_", None, pi, pi)
        self.f_escprob = self._fn("_escprob_", C.c_double, pd)
        self.f_backrad = self._fn("_backrad_", None)
        self.f_lubksb = self._fn("_lubksb_", None, pd, pi, pi, pi, pd)
        self.f_readdata = self._fn("_readdata_", None)

    def common_f64(self, name, byte_off, n):
        return np.ctypeslib.as_array((C.c_double * n).from_address(self.addr(name) + byte_off))

    def common_i32(self, name, byte_off, n):
        return np.ctypeslib.as_array((C.c_int32 * n).from_address(self.addr(name) + byte_off))

    # --- COMMON layout, SURVEY.md Appendix B ----------------------------------
    def views(self):
        v = {}
        v["imolec_hdr"] = self.common_i32("_imolec_", 0, 5)      # nlev,nline,ncoll,npart,ntemp
        v["iupp"] = self.common_i32("_imolec_", 0x14, MAXLINE)
        v["ilow"] = self.common_i32("_imolec_", 0x61A90, MAXLINE)
        v["amass"] = self.common_f64("_rmolec_", 0, 1)
        v["eterm"] = self.common_f64("_rmolec_", 0x8, MAXLEV)
        v["gstat"] = self.common_f64("_rmolec_", 0x5DC0, MAXLEV)
        v["aeinst"] = self.common_f64("_rmolec_", 0xBB78, MAXLINE)
        v["density"] = self.common_f64("_cphys_", 0, 9)
        v["tkin"] = self.common_f64("_cphys_", 0x48, 1)
        v["tbg"] = self.common_f64("_cphys_", 0x50, 1)
        v["cdmol"] = self.common_f64("_cphys_", 0x58, 1)
        v["deltav"] = self.common_f64("_cphys_", 0x60, 1)
        v["totdens"] = self.common_f64("_cphys_", 0x68, 1)
        v["xnu"] = self.common_f64("_radi_", 0, MAXLINE)
        v["taul"] = self.common_f64("_radi_", 0xC34F8, MAXLINE)
        v["tex"] = self.common_f64("_radi_", 0x1869F0, MAXLINE)
        v["backi"] = self.common_f64("_radi_", 0x249EE8, MAXLINE)
        v["totalb"] = self.common_f64("_radi_", 0x30D3E0, MAXLINE)
        v["spfreq"] = self.common_f64("_radi_", 0x3D08D8, MAXLINE)
        v["trj"] = self.common_f64("_radi_", 0x493DD0, MAXLINE)
        v["crate"] = self.common_f64("_collie_", 0, MAXLEV * MAXLEV)   # column-major (i,j)
        v["ctot"] = self.common_f64("_collie_", 0x449E688, MAXLEV)
        v["xpop"] = self.common_f64("_collie_", 0x44A4440, MAXLEV)
        v["method"] = self.common_i32("_setup_", 0x78, 1)
        return v

    # --- the four routines -------------------------------------------------------
    def escprob(self, tau: float, method: int) -> float:
        self.views()["method"][0] = method
        t = C.c_double(tau)
        return self.f_escprob(C.byref(t))

    def matrix(self, niter: int, conv: int) -> int:
        a, b = C.c_int(niter), C.c_int(conv)
        self.f_matrix(C.byref(a), C.byref(b))
        return b.value

    def backrad(self):
        self.f_backrad()

    def readdata(self, molfile: str, tkin: float, density_by_id: dict):
        """The reference's own readdata_ on `molfile` (impex.molfile, 120 characters) with cphys.tkin and
        cphys.density(id) set as the setters do (emcee/pyradex/core.py:401-402, 489-579): parses the LAMDA
        file, fills imolec / rmolec / radi.xnu, interpolates the rates, detailed balance, ctot."""
        if len(molfile) > 120:
            raise ValueError("molfile path longer than the COMMON block's 120 characters")
        C.memmove(self.addr("_impex_") + 0x78, molfile.ljust(120).encode(), 120)
        v = self.views()
        if getattr(self, "_molfile", None) not in (None, molfile):
            # readdata_ keeps the interpolated rates of a partner in a static table that it only ever writes where the
            # file lists a transition, and adds density * table into crate for every pair of levels: entries of ANOTHER
            # molecule read earlier in the same image would leak in (seen: toy6 after co_synth).  The reference's
            # process reads one molecule, so its unlisted pairs hold the zeros of a fresh image: one image per molecule.
            raise ValueError("this image has read %s: map a fresh RefRadex for another molecule" % self._molfile)
        self._molfile = molfile
        v["density"][:] = 0.0
        for k, x in density_by_id.items():
            v["density"][int(k) - 1] = x
        v["totdens"][0] = float(sum(density_by_id.values()))
        v["tkin"][0] = tkin
        self.f_readdata()

    def lubksb(self, a_colmajor: np.ndarray):
        """a: (np_, np_) Fortran-ordered; n = np_ (the reference passes nplus, maxlev)."""
        npd = a_colmajor.shape[0]
        A = np.asfortranarray(a_colmajor, dtype=np.float64).copy(order="F")
        b = np.zeros(npd)
        indx = np.zeros(npd, dtype=np.int32)
        n, np_ = C.c_int(npd), C.c_int(npd)
        self.f_lubksb(A.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n), C.byref(np_),
                      indx.ctypes.data_as(C.POINTER(C.c_int)),
                      b.ctypes.data_as(C.POINTER(C.c_double)))
        return b
