"""Code-object audit of the two kernels that manage the accumulator file by hand (ode_bf16x6w.hip, gemm_bf16x6w.hip).

Both keep 256 live values at FIXED addresses a0..a255 across separate inline-asm statements.  A clobber list does not reserve
registers between statements: what keeps hipcc out of the AGPRs is the hidden flag -amdgpu-mfma-vgpr-form (build.py EXTRA) plus the
fact that a kernel without spills has no reason to park anything there.  Neither is a contract, so the BUILD checks the result:
build.build() runs audit_objects() on the objects it links and refuses to produce the library when the check trips (a silently
corrupted headline kernel is worse than no library).  tests/test_host_cpu.py runs the same functions on the in-tree objects.

TESTED_HIPCC is the compiler the committed expectations were measured with; another version is not an error in itself (the audit
decides), but the message of a failed audit names both."""
import os
import re
import subprocess
import tempfile

LLVM = os.environ.get("CASPR_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TESTED_HIPCC = "HIP version: 7.2.26015-fc0010cf6a / AMD clang version 22.0.0git roc-7.2.0"


class AuditError(RuntimeError):
    pass


def tools_present():
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objdump", "llvm-objcopy", "llvm-readelf", "clang-offload-bundler"))


def _code_object(obj):
    """-> (notes text, disassembly text) of the gfx950 code object bundled in a hipcc -c object."""
    with tempfile.TemporaryDirectory() as d:
        fat, elf = os.path.join(d, "w.fatbin"), os.path.join(d, "w.elf")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(d, "copy.o")])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf])
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", elf], text=True)
        dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", elf], text=True)
    return notes, dis


def _kernel(notes, dis, name):
    """-> (the metadata entry of kernel `name` in the amdhsa.kernels note, its instruction list).  The note is a YAML list whose
    entries start with "  - ." at list depth; the entry is found by its `.name:` line, not by character windows."""
    entry, cur = None, []
    for ln in notes.splitlines() + ["  - .end"]:
        if re.match(r"^\s{0,4}- \.", ln):
            if any(re.match(r"^\s*(- )?\.name:\s+%s\s*$" % re.escape(name), l) for l in cur):
                entry = "\n".join(cur)
                break
            cur = [ln]
        else:
            cur.append(ln)
    if entry is None:
        raise AuditError("%s: no metadata entry in the code object's notes" % name)
    try:
        body = dis[dis.index("<%s>:" % name):]
        body = body[:body.index("s_endpgm")]
    except ValueError:
        raise AuditError("%s: not found in the disassembly" % name) from None
    ins = [ln.split("//")[0].strip() for ln in body.splitlines() if ln.startswith("\t")]
    return entry, ins


def _need(cond, what):
    if not cond:
        raise AuditError(what)


def _no_spills(meta, ins, k):
    _need(re.search(r"\.private_segment_fixed_size:\s+0\b", meta) and re.search(r"\.vgpr_spill_count:\s+0\b", meta), "%s: scratch / spills in the metadata" % k)
    _need(not any(i.startswith("scratch_") for i in ins), "%s: scratch instructions" % k)


def audit_cnf_x6w(obj):
    """cnf_rk4_x6w_kernel: no scratch, exactly the accumulator moves the source writes (256 zeroing writes + 256 in-place activation
    writes, 256 + 256 reads in pass 0 / passes 1-3), layer 1's MFMAs on a[..], every other MFMA in the VGPR form, M0 written once per
    LDS-DMA statement and by nothing else."""
    notes, dis = _code_object(obj)
    k = "_Z18cnf_rk4_x6w_kernel9CnfX6Args"
    meta, ins = _kernel(notes, dis, k)
    count = lambda pat: sum(1 for i in ins if re.match(pat, i))
    _no_spills(meta, ins, k)
    _need(re.search(r"\.agpr_count:\s+256\b", meta), k + ": agpr_count != 256")
    r, w = count(r"v_accvgpr_read_b32"), count(r"v_accvgpr_write_b32")
    _need(r == 512 and w == 512 and count(r"v_accvgpr_mov") == 0, "%s: %d reads / %d writes of the accumulator file (512 / 512 in the source): the compiler touches it" % (k, r, w))
    mfma = [i for i in ins if i.startswith("v_mfma")]
    on_acc = [i for i in mfma if re.match(r"v_mfma_f32_32x32x16_bf16 a\[", i)]
    _need(len(on_acc) == 8 * 4 * 12 and all(" a[" not in i for i in mfma if i not in on_acc), "%s: %d of %d MFMAs on a[..] (384 expected, all others on VGPRs)" % (k, len(on_acc), len(mfma)))
    dma = count(r"global_load_lds_dwordx4")
    m0 = sum(1 for i in ins if re.search(r"\bm0\b", i))
    _need(dma > 20 and m0 == dma - 10, "%s: %d LDS-DMA instructions, %d M0 accesses" % (k, dma, m0))
    return {"kernel": k, "accvgpr_reads": r, "accvgpr_writes": w, "mfma": len(mfma), "mfma_on_acc": len(on_acc), "lds_dma": dma}


def audit_conv_x6w(obj):
    """conv1x1_x6w_kernel (all instantiations; persistent since round 4): no scratch, every MFMA on the hand-managed a[..] tiles, the
    accumulator file zeroed once in the prologue and read out + zeroed again in the two instances of the tile read-out (one per chunk
    parity): 3 x 256 writes, 2 x 256 reads, and nothing else."""
    notes, dis = _code_object(obj)
    kernels = re.findall(r"<(_Z18conv1x1_x6w_kernelI[^>]*)>:", dis)
    _need(len(kernels) >= 4, "conv1x1_x6w_kernel: %d instantiations found" % len(kernels))
    out = []
    for k in kernels:
        meta, ins = _kernel(notes, dis, k)
        count = lambda pat: sum(1 for i in ins if re.match(pat, i))
        _no_spills(meta, ins, k)
        r, w = count(r"v_accvgpr_read_b32"), count(r"v_accvgpr_write_b32")
        _need(w == 768 and r == 512 and count(r"v_accvgpr_mov") == 0, "%s: %d reads / %d writes of the accumulator file (512 / 768 in the source)" % (k, r, w))
        mfma = [i for i in ins if i.startswith("v_mfma")]
        _need(len(mfma) == 2 * 192 and all(re.match(r"v_mfma_f32_32x32x16_bf16 a\[", i) for i in mfma), "%s: %d MFMAs, not all on a[..]" % (k, len(mfma)))
        out.append({"kernel": k, "accvgpr_reads": r, "accvgpr_writes": w, "mfma": len(mfma)})
    return out


AUDITS = {"ode_bf16x6w.hip": audit_cnf_x6w, "gemm_bf16x6w.hip": audit_conv_x6w}


def audit_objects(objs_by_source):
    """objs_by_source: {source file name: object path}.  Raises AuditError (naming the compiler) on the first violation -- including
    missing LLVM tools and anything unexpected while taking the code object apart.  CASPR_SKIP_AUDIT=1 (build.py) turns a failure
    into a loud warning: for a ROCm point release that schedules differently but correctly -- after `pytest -m gpu` has shown that
    the two kernels still compute what the oracle says."""
    if not tools_present():
        raise AuditError("the ROCm LLVM tools (%s: llvm-objdump, llvm-objcopy, llvm-readelf, clang-offload-bundler) are needed to audit the "
                         "hand-managed accumulator kernels" % LLVM)
    res = {}
    for src, fn in AUDITS.items():
        if src in objs_by_source:
            try:
                res[src] = fn(objs_by_source[src])
            except (AuditError, ValueError, IndexError, subprocess.CalledProcessError) as e:
                ver = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--version"], capture_output=True, text=True).stdout.splitlines()[:2]
                raise AuditError("%s\n  this compiler: %s\n  tested with:   %s\n  -> the library is NOT produced (CASPR_SKIP_AUDIT=1 overrides, "
                                 "see caspr_amd/csrc/audit.py)" % (e, " / ".join(ver), TESTED_HIPCC)) from None
    return res


def audit_no_packed_f32(objs_by_source):
    """No v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 in any product code object (build.py: NO_PACKED_F32).  Round 6: a packed-f32 operation that
    consumes a register an LDS read has just returned was seen to use the register's OLD content in one 16-lane pass when another kernel that also
    executes packed-f32 instructions shares the compute unit (tools/micro/pk_check.hip, profiles/r06_pk_check.txt); the library is built without the
    instruction class, and this check keeps a changed flag set or compiler from bringing it back unnoticed.  -> {source: 0}."""
    if not tools_present():
        raise AuditError("the ROCm LLVM tools (%s) are needed to check the objects for packed-f32 instructions" % LLVM)
    res = {}
    for src, obj in sorted(objs_by_source.items()):
        try:
            _, dis = _code_object(obj)
        except subprocess.CalledProcessError as e:
            raise AuditError("%s: could not take the code object apart (%s)" % (src, e)) from None
        hits = re.findall(r"\bv_pk_(?:add|mul|fma)_f32\b", dis)
        if hits:
            raise AuditError("%s: %d packed-f32 VALU instructions in the gfx950 code object (build.py compiles with -target-feature -packed-fp32-ops; see "
                             "audit_no_packed_f32)" % (src, len(hits)))
        res[src] = 0
    return res
