#!/usr/bin/env python3
"""include/vpbs_prover.h -> bindings/rust/vpbs_sys.rs: the Rust `extern "C"` mirror of the WHOLE C ABI, generated, so that the binding a
maintainer of the reference (/root/reference, Rust; call sites src/vtfhe/ivc_based_vpbs.rs:302,333,364,446,488) pastes into the patched
plonky2 cannot drift from the header.  No Rust toolchain exists in the authoring image: the file is checked by
tests/test_host_cpu.py::test_rust_binding_matches_the_header (regenerated text == committed text; every declaration INTEGRATION.md shows is
one of its lines) instead of by rustc.

usage: tools/gen_rust_ffi.py [--check] [--integration]   (writes bindings/rust/vpbs_sys.rs, or with --check exits 1 when it is stale;
       --integration also rewrites the excerpt INTEGRATION.md section 2 shows, between its BEGIN / END markers)

The parser handles exactly the C subset the header uses: opaque struct typedefs, plain structs (scalar / pointer / fixed-array fields,
several declarators per line), enums, function-pointer typedefs and prototypes."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vpbs_prover.h")
OUT = os.path.join(ROOT, "bindings", "rust", "vpbs_sys.rs")

SCALARS = {"int": "i32", "unsigned": "u32", "unsigned int": "u32", "long": "c_long", "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32",
           "uint8_t": "u8", "double": "f64", "char": "c_char", "void": "c_void"}
RUST_KEYWORDS = {"in", "fn", "type", "ref", "loop", "match", "move", "box", "use", "where", "as", "mod", "self", "super", "crate", "impl"}


def camel(name):
    return "".join(p.capitalize() for p in name.split("_"))


def strip(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    lines = [ln for ln in text.splitlines() if not ln.lstrip().startswith("#")]
    text = "\n".join(lines)
    text = text.replace('extern "C" {', " ")
    return text


def statements(text):
    """top-level statements (split at ';' outside braces); a stray closing brace of the extern block is dropped"""
    out, depth, cur = [], 0, []
    for ch in text:
        if ch == "{":
            depth += 1
        elif ch == "}":
            if depth == 0:
                continue
            depth -= 1
        if ch == ";" and depth == 0:
            s = " ".join("".join(cur).split())
            if s:
                out.append(s)
            cur = []
        else:
            cur.append(ch)
    return out


class Binding:
    def __init__(self, header_text):
        self.opaque, self.structs, self.enums, self.fnptrs, self.functions = [], [], [], [], []
        self.known = {}
        for s in statements(strip(header_text)):
            self.statement(s)

    # ---- types ----
    def base_type(self, c):
        c = c.strip()
        if c in SCALARS:
            return SCALARS[c]
        if c in self.known:
            return self.known[c]
        raise ValueError("unknown C type %r" % c)

    def rust_type(self, ctype, array=None, param=False):
        """ctype: e.g. 'const uint64_t*', 'vpbs_batch* const*', 'void*'; array: the [..] suffix of the declarator (None, '', '12')"""
        t = ctype.strip()
        # split off pointer levels from the right: each '*' optionally preceded by 'const' qualifying the POINTER to its left
        levels = []   # constness of the pointee at each level, innermost first
        m = re.match(r"^(const\s+)?([A-Za-z_][A-Za-z0-9_ ]*?)\s*((?:\*\s*(?:const\s*)?)*)$", t)
        if not m:
            raise ValueError("cannot parse type %r" % ctype)
        inner_const, base, stars = bool(m.group(1)), m.group(2).strip(), m.group(3)
        ptrs = re.findall(r"\*\s*(const)?", stars)
        r = self.base_type(base)
        pointee_const = inner_const
        for q in ptrs:
            if base in self.fn_names() and not levels:
                pass
            r = ("*const " if pointee_const else "*mut ") + r
            pointee_const = q == "const"
            levels.append(q)
        if array is not None:
            if param:   # an array parameter decays to a pointer to its element
                r = ("*const " if inner_const and not ptrs else "*mut ") + r if not ptrs else r
                if ptrs:
                    raise ValueError("array of pointers as a parameter is not used by the header")
            else:
                r = "[%s; %s]" % (r, array)
        if base in self.fn_names() and not ptrs:
            r = "Option<%s>" % self.known[base]
        return r

    def fn_names(self):
        return {n for n, _, _ in self.fnptrs}

    # ---- declarations ----
    def params(self, text):
        text = text.strip()
        if text in ("", "void"):
            return []
        out = []
        for i, p in enumerate(self.split_commas(text)):
            m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*(\[\s*(\w*)\s*\])?$", p.strip())
            if not m:
                raise ValueError("cannot parse parameter %r" % p)
            ctype, name, arr = m.group(1), m.group(2), m.group(3)
            if not ctype.strip():   # unnamed parameter: the "name" was the type
                ctype, name = name, "arg%d" % i
            out.append((self.rust_name(name), self.rust_type(ctype, m.group(4) if arr else None, param=True)))
        return out

    @staticmethod
    def rust_name(n):
        n = n.lower() if n.isupper() else n
        n = re.sub(r"([a-z])([A-Z])", lambda m: m.group(1) + "_" + m.group(2).lower(), n).lower()
        return n + "_" if n in RUST_KEYWORDS else n

    @staticmethod
    def split_commas(text):
        out, depth, cur = [], 0, []
        for ch in text:
            if ch in "([":
                depth += 1
            elif ch in ")]":
                depth -= 1
            if ch == "," and depth == 0:
                out.append("".join(cur))
                cur = []
            else:
                cur.append(ch)
        out.append("".join(cur))
        return out

    def statement(self, s):
        m = re.match(r"^typedef struct (\w+) (\w+)$", s)
        if m:
            self.known[m.group(2)] = camel(m.group(2))
            self.opaque.append(m.group(2))
            return
        m = re.match(r"^typedef struct \{(.*)\} (\w+)$", s)
        if m:
            self.known[m.group(2)] = camel(m.group(2))
            fields = []
            for decl in statements(m.group(1) + ";"):
                dm = re.match(r"^((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*(?:\s+int)?\s*\**)\s*(.*)$", decl)
                ctype, rest = dm.group(1), dm.group(2)
                for d in self.split_commas(rest):
                    d = d.strip()
                    extra = re.match(r"^(\**)\s*(\w+)\s*(\[\s*(\w+)\s*\])?$", d)
                    fields.append((self.rust_name(extra.group(2)), self.rust_type(ctype + extra.group(1), extra.group(4) if extra.group(3) else None)))
            self.structs.append((m.group(2), fields))
            return
        m = re.match(r"^typedef enum \{(.*)\} (\w+)$", s)
        if m:
            self.known[m.group(2)] = "i32"
            items, value = [], -1
            for it in self.split_commas(m.group(1)):
                it = it.strip()
                if not it:
                    continue
                if "=" in it:
                    name, v = (x.strip() for x in it.split("="))
                    value = int(v, 0)
                else:
                    name, value = it, value + 1
                items.append((name, value))
            self.enums.append((m.group(2), items))
            return
        m = re.match(r"^typedef (.*?)\(\s*\*\s*(\w+)\s*\)\s*\((.*)\)$", s)
        if m:
            ret, name, args = m.group(1).strip(), m.group(2), m.group(3)
            ps = self.params(args)
            sig = "unsafe extern \"C\" fn(%s)" % ", ".join("%s: %s" % p for p in ps)
            if ret != "void":
                sig += " -> " + self.rust_type(ret)
            self.known[name] = camel(name)
            self.fnptrs.append((name, sig, ps))
            return
        m = re.match(r"^(.*?)\b(vpbs_\w+)\s*\((.*)\)$", s)
        if m:
            ret, name, args = m.group(1).strip(), m.group(2), m.group(3)
            self.functions.append((name, self.params(args), None if ret == "void" else self.rust_type(ret)))
            return
        raise ValueError("unrecognised statement: " + s[:120])

    # ---- output ----
    def fn_line(self, name, ps, ret):
        return "    pub fn %s(%s)%s;" % (name, ", ".join("%s: %s" % p for p in ps), " -> " + ret if ret else "")

    def render(self):
        o = ["// GENERATED by tools/gen_rust_ffi.py from include/vpbs_prover.h -- do not edit; regenerate after every change of the header.",
             "// The C ABI of libvpbs_hip.so for the Rust side of /root/reference (the patched plonky2 behind prove(), ivc_based_vpbs.rs:302,333,364):",
             "// link with  println!(\"cargo:rustc-link-lib=dylib=vpbs_hip\")  (INTEGRATION.md section 1).  Semantics of every entry point: the header.",
             "#![allow(non_camel_case_types, dead_code)]",
             "use std::os::raw::{c_char, c_long, c_void};", "",
             "pub const VPBS_POW_ANY: u64 = u64::MAX;", "pub const VPBS_UNUSED_SELECTOR: u32 = 0xFFFF_FFFF;", ""]
        for n in self.opaque:
            o.append("#[repr(C)] pub struct %s { _private: [u8; 0] }" % camel(n))
        o.append("")
        for n, items in self.enums:
            o.append("// %s" % n)
            for name, v in items:
                o.append("pub const %s: i32 = %d;" % (name, v))
            o.append("")
        for n, sig, _ in self.fnptrs:
            o.append("pub type %s = %s;" % (camel(n), sig))
        o.append("")
        for n, fields in self.structs:
            o.append("#[repr(C)] #[derive(Clone, Copy)]")
            o.append("pub struct %s {" % camel(n))
            for f, t in fields:
                o.append("    pub %s: %s," % (f, t))
            o.append("}")
            o.append("")
        o.append("extern \"C\" {")
        for name, ps, ret in self.functions:
            o.append(self.fn_line(name, ps, ret))
        o.append("}")
        return "\n".join(o) + "\n"


def generate():
    return Binding(open(HEADER).read()).render()


# the declarations INTEGRATION.md section 2 shows (verbatim lines of the generated binding), by group
EXCERPT = [
    ("context, compatibility table, errors, host settings",
     ["vpbs_ctx_create", "vpbs_ctx_destroy", "vpbs_last_error", "vpbs_compat_default", "vpbs_ctx_set_compat", "vpbs_ctx_set_option", "vpbs_host_set_cpu_budget",
      "vpbs_host_set_late_threads", "vpbs_host_set_early_threads", "vpbs_host_set_blocking_sync", "vpbs_host_set_sync_word", "vpbs_hash_no_pad", "vpbs_hash_pad", "vpbs_hash_chain",
      "vpbs_hash_chain_links", "vpbs_circuit_digest"]),
    ("PolynomialBatch / OpeningSet / prove_openings (fri/oracle.rs): the seam inside the patched plonky2",
     ["vpbs_commit_values", "vpbs_commit_coeffs", "vpbs_batch_free", "vpbs_batch_lde_rows", "vpbs_batch_eval_ext", "vpbs_batch_open", "vpbs_fri_proof_words",
      "vpbs_fri_prove"]),
    ("permutation argument + quotient stage (device-resident: the batches never leave HBM)",
     ["vpbs_partial_products", "vpbs_gates_layout", "vpbs_gate_terms", "vpbs_quotient_permutation"]),
    ("the whole of prove() after witness generation, its serialiser, cd.verify (ivc_based_vpbs.rs:302,333,364 / :488 / :446)",
     ["vpbs_step_sizes_get", "vpbs_prove_step", "vpbs_prove_step_sharded", "vpbs_prove_step_sharded_fail", "vpbs_comm_allgather_checked",
      "vpbs_step_proof_to_bytes", "vpbs_step_proof_from_bytes", "vpbs_verify_step"]),
    ("witness generation: compiled once per circuit; two-phase (late phase in stages) for a chain whose PartialWitness ends with the previous proof",
     ["vpbs_witness_plan_create", "vpbs_witness_plan_run", "vpbs_witness_plan_split", "vpbs_witness_plan_run_early", "vpbs_witness_plan_late_stages",
      "vpbs_witness_plan_run_late_stage", "vpbs_witness_plan_run_late_packed", "vpbs_witness_plan_late_count", "vpbs_witness_plan_late_positions",
      "vpbs_device_scatter", "vpbs_witness_device_create", "vpbs_witness_device_run", "vpbs_witness_device_wires", "vpbs_witness_device_read"]),
    ("one verifiable PBS as verified_pbs / verify_pbs see it (ivc_based_vpbs.rs:159-386, :388-489)",
     ["vpbs_ivc_create", "vpbs_ivc_set_step_callback", "vpbs_ivc_set_device_witness", "vpbs_ivc_last_error", "vpbs_ivc_verifier_data", "vpbs_ivc_prove_pbs",
      "vpbs_ivc_free", "vpbs_verify_pbs"]),
    ("RCCL collectives of a sharded step (the library binds librccl.so itself)",
     ["vpbs_rccl_available", "vpbs_rccl_unique_id", "vpbs_comm_rccl_create", "vpbs_comm_rccl_destroy"]),
]
BEGIN, END = "// BEGIN excerpt of bindings/rust/vpbs_sys.rs (tools/gen_rust_ffi.py --integration)", "// END excerpt"


def integration_excerpt():
    b = Binding(open(HEADER).read())
    fn = {n: (ps, ret) for n, ps, ret in b.functions}
    lines = [BEGIN, "extern \"C\" {"]
    for title, names in EXCERPT:
        lines.append("    // " + title)
        lines += [b.fn_line(n, *fn[n]) for n in names]
    lines += ["}", END]
    return "\n".join(lines)


def rewrite_integration():
    path = os.path.join(ROOT, "INTEGRATION.md")
    doc = open(path).read()
    a, z = doc.index(BEGIN), doc.index(END) + len(END)
    new = doc[:a] + integration_excerpt() + doc[z:]
    if new != doc:
        open(path, "w").write(new)
    return new != doc


if __name__ == "__main__":
    text = generate()
    if "--check" in sys.argv:
        stale = not os.path.exists(OUT) or open(OUT).read() != text
        print("stale" if stale else "up to date")
        sys.exit(1 if stale else 0)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    if "--integration" in sys.argv:
        print("INTEGRATION.md excerpt %s" % ("rewritten" if rewrite_integration() else "up to date"))
    b = Binding(open(HEADER).read())
    print("%s: %d functions, %d structs, %d opaque types, %d enums, %d callback types" %
          (os.path.relpath(OUT, ROOT), len(b.functions), len(b.structs), len(b.opaque), len(b.enums), len(b.fnptrs)))
