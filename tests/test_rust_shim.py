"""CPU: the Rust shim (shim/perceive-core/) against the C header, mechanically — the image has no Rust
toolchain, so the shim is source only.  Checked: `src/ffi.rs` declares exactly the header's symbols with the
same argument counts, pointer shapes and integer widths, the same struct fields and enum values, and is what
tools/gen_rust_ffi.py produces today; the wrapper modules only call functions that exist and define every
public item of the reference that out-of-crate callers use."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "shim", "perceive-core")
HEADER = os.path.join(ROOT, "include", "perceive_hip.h")

C_SCALAR = {"pcv_status": "i32", "int": "i32", "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64",
            "size_t": "usize", "float": "f32", "double": "f64", "uint8_t": "u8", "char": "i8", "void": "void"}
R_SCALAR = {"c_int": "i32", "i32": "i32", "u32": "u32", "i64": "i64", "u64": "u64", "usize": "usize", "f32": "f32", "f64": "f64",
            "u8": "u8", "c_char": "i8", "c_void": "void"}


def c_canon(t):
    """'const int64_t*' -> ('i64', ['const']); 'pcv_ctx**' -> ('pcv_ctx', ['mut', 'mut'])  (pointer levels, innermost first)"""
    t = t.strip()
    if re.search(r"\[\d*\]$", t):
        t = re.sub(r"\[\d*\]$", "*", t)
    toks = re.findall(r"const|\*|\w+", t)
    base = next(x for x in toks if x not in ("const", "*"))
    levels, const = [], False
    seen_base = False
    for x in toks:
        if x == base and not seen_base:
            seen_base = True
        elif x == "const":
            const = True
        elif x == "*":
            levels.append("const" if const else "mut")
            const = False
    return C_SCALAR.get(base, base), levels


def r_canon(t):
    t = t.strip()
    levels = []
    while t.startswith("*"):
        m = re.match(r"\*(const|mut)\s+(.*)", t)
        levels.append(m.group(1))
        t = m.group(2)
    levels.reverse()  # innermost first
    return R_SCALAR.get(t, t), levels


def header_functions():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    text = re.sub(r"typedef struct \w+ \{.*?\} \w+;", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"([\w \*]+?)\b(pcv_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        plist = []
        if params not in ("", "void"):
            for p in params.split(","):
                pm = re.match(r"(.*?)(\w+)\s*(\[\d*\])?$", p.strip())
                plist.append(c_canon(pm.group(1) + (pm.group(3) or "")))
        out[name] = (c_canon(ret), plist)
    return out


def rust_functions():
    text = open(os.path.join(SHIM, "src", "ffi.rs")).read()
    block = text[text.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (pcv_[a-z0-9_]+)\((.*?)\)(?:\s*->\s*([^;]+))?;", block):
        name, params, ret = m.group(1), m.group(2), m.group(3)
        plist = [r_canon(p.split(":", 1)[1]) for p in params.split(",") if p.strip()]
        out[name] = (r_canon(ret) if ret else ("void", []), plist)
    return out


def test_ffi_matches_header_symbol_for_symbol():
    c, r = header_functions(), rust_functions()
    assert len(c) >= 70
    assert sorted(c) == sorted(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name in c:
        cret, cargs = c[name]
        rret, rargs = r[name]
        assert cret == rret, (name, cret, rret)
        assert len(cargs) == len(rargs), (name, len(cargs), len(rargs))
        for i, (ca, ra) in enumerate(zip(cargs, rargs)):
            assert ca == ra, (name, i, ca, ra)  # same scalar width / struct, same pointer depth and constness


def test_ffi_symbols_are_exported_by_the_library():
    from perceive_amd import _ffi

    out = subprocess.run(["nm", "-D", "--defined-only", _ffi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (pcv_[a-z0-9_]+)", out))
    assert set(rust_functions()) <= exported


def test_ffi_structs_and_constants():
    htext = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    rtext = open(os.path.join(SHIM, "src", "ffi.rs")).read()
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} \w+;", htext, flags=re.S):
        name, body = m.group(1), m.group(2)
        cfields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                ctype, names = decl.split(None, 1)
                cfields += [(n.strip(), C_SCALAR[ctype]) for n in names.split(",")]
        rm = re.search(r"pub struct %s \{(.*?)\}" % name, rtext, flags=re.S)
        assert rm and "#[repr(C)]" in rtext[: rm.start()].rsplit("\n\n", 1)[-1], name
        rfields = [(f.group(1), R_SCALAR[f.group(2)]) for f in re.finditer(r"pub (\w+): (\w+),", rm.group(1))]
        assert cfields == rfields, name
    consts = {}
    for m in re.finditer(r"enum\s*\{(.*?)\};", htext, flags=re.S):
        for item in m.group(1).split(","):
            if item.strip():
                k, v = [x.strip() for x in item.split("=")]
                consts[k] = int(v)
    rconsts = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (PCV_\w+): c_int = (-?\d+);", rtext)}
    assert consts == rconsts and len(consts) >= 20


def test_ffi_callback_types_match_the_header():
    htext = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    rtext = open(os.path.join(SHIM, "src", "ffi.rs")).read()
    found = 0
    for m in re.finditer(r"typedef\s+([\w \*]+?)\(\s*\*\s*(pcv_\w+)\s*\)\s*\(([^;]*?)\)\s*;", htext, flags=re.S):
        ret, name, params = m.group(1), m.group(2), " ".join(m.group(3).split())
        cargs = []
        for p in params.split(","):
            pm = re.match(r"(.*?)(\w+)\s*$", p.strip())
            cargs.append(c_canon(pm.group(1)))
        rm = re.search(r'pub type %s = Option<unsafe extern "C" fn\((.*?)\)(?:\s*->\s*([^>;]+))?>;' % name, rtext)
        assert rm, name
        rargs = [r_canon(p.split(":", 1)[1]) for p in rm.group(1).split(",") if p.strip()]
        assert cargs == rargs, (name, cargs, rargs)
        assert c_canon(ret) == (r_canon(rm.group(2)) if rm.group(2) else ("void", []))
        found += 1
    assert found >= 1


def test_ffi_is_what_the_generator_writes():
    import importlib.util
    import shutil
    import tempfile

    spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(ROOT, "tools", "gen_rust_ffi.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    with tempfile.TemporaryDirectory() as tmp:
        gen.OUT = os.path.join(tmp, "ffi.rs")
        gen.main()
        assert open(gen.OUT).read() == open(os.path.join(SHIM, "src", "ffi.rs")).read(), "run tools/gen_rust_ffi.py"


def _code(path):
    """source with comments and string literals blanked (for brace counting and identifier searches)"""
    s = open(path).read()
    s = re.sub(r'r##".*?"##', '""', s, flags=re.S)
    s = re.sub(r'"(?:\\.|[^"\\])*"', '""', s, flags=re.S)
    s = re.sub(r"//[^\n]*", "", s)
    return s


def test_wrappers_call_only_declared_functions_and_are_balanced():
    declared = set(rust_functions())
    used = set()
    for f in ("hip.rs", "model.rs", "search.rs", "configs.rs", "similarity.rs"):
        code = _code(os.path.join(SHIM, "src", f))
        for a, b in ("{}", "()", "[]"):
            assert code.count(a) == code.count(b), (f, a, code.count(a), code.count(b))
        used |= set(re.findall(r"ffi::(pcv_[a-z0-9_]+)\s*\(", code))  # calls (ffi::pcv_ctx etc. are the handle types)
    assert used and used <= declared, sorted(used - declared)
    # the calls the hot path needs are there
    for need in ("pcv_init", "pcv_model_create_from_dir", "pcv_model_encode_text", "pcv_model_highlight", "pcv_model_destroy",
                 "pcv_searcher_create", "pcv_searcher_add_blobs", "pcv_searcher_clear_source", "pcv_searcher_finalize",
                 "pcv_searcher_search", "pcv_searcher_destroy", "pcv_serialize_embedding", "pcv_deserialize_embedding",
                 "pcv_searcher_replace_source", "pcv_searcher_dim", "pcv_dot_product", "pcv_cosine_similarity"):
        assert need in used, need


def test_crate_level_similarity_functions_and_documented_signature_change():
    """lib.rs:63-77 (dot_product, cosine_similarity_{single,multi}_query) exist over the C ABI, and the one public
    signature that changes — new_pretrained's error type (model.rs:68) — is written down with the call sites it touches."""
    sim = open(os.path.join(SHIM, "src", "similarity.rs")).read()
    for pat in (r"pub fn dot_product\(set1: &Embeddings, set2: &Embeddings\) -> Embeddings",
                r"pub fn cosine_similarity_single_query\(query: &Embeddings, matches: &Embeddings\) -> Embeddings",
                r"pub fn cosine_similarity_multi_query\(set1: &Embeddings, set2: &Embeddings\) -> Embeddings"):
        assert re.search(pat, sim), pat
    readme = open(os.path.join(SHIM, "README.md")).read()
    for need in ("RustBertError", "ModelError", "perceive-cli/state.rs:46-48", "perceive-tauri/src-tauri/main.rs:76", "similarity.rs"):
        assert need in readme, need
    # rebuild_source stages the replacement and swaps (search.rs:57-79); search_vector checks the query width before the FFI call
    search = _code(os.path.join(SHIM, "src", "search.rs"))
    body = search[search.index("pub fn rebuild_source"):search.index("fn load_sources")]
    assert body.index("PCV_STAGING_SOURCE") < body.index("pcv_searcher_replace_source") and "pcv_searcher_clear_source(self.handle, source_id)" not in body
    sv = search[search.index("pub fn search_vector("):search.index("pub fn search(")]
    assert sv.index("pcv_searcher_dim") < sv.index("pcv_searcher_search(") and "assert_eq!(vector.len()" in sv and ".min(128)" not in sv


def test_public_surface_of_the_reference_is_kept():
    model = open(os.path.join(SHIM, "src", "model.rs")).read()
    search = open(os.path.join(SHIM, "src", "search.rs")).read()
    configs = open(os.path.join(SHIM, "src", "configs.rs")).read()
    for pat in (r"pub struct Model \{\s*pub model_type: SentenceEmbeddingsModelType",            # model.rs:56-57
                r"pub fn new_pretrained\(model_type: SentenceEmbeddingsModelType\) -> Result<Model, ",  # model.rs:68
                r"pub fn encode<S: AsRef<str> \+ Sync>\(&self, inputs: &\[S\]\) -> Result<",     # model.rs:176
                r"impl From<Embeddings> for Vec<Vec<f32>>",                                       # calculate_embeddings.rs:21
                r"pub fn highlight<'s, 'doc, S: AsRef<str> \+ Sync>\(\s*&'s self,\s*query: &'doc str,\s*documents: &'doc \[S\],\s*\) -> Result<Vec<Option<&'doc str>>, ModelError>",
                r"unsafe impl Send for Model", r"unsafe impl Sync for Model", r"pub enum ModelError", r"pub use configs::SentenceEmbeddingsModelType"):
        assert re.search(pat, model), pat
    for pat in (r"#\[derive\(Debug, Copy, Clone\)\]\s*pub struct SearchItem \{\s*pub id: i64,\s*pub score: f32,",   # search.rs:18-22
                r"pub hidden: HashSet<i64>",                                                                        # search.rs:34
                r"pub fn build\(database: &Database, model_id: u32, model_version: u32\) -> Result<Searcher, eyre::Report>",
                r"pub fn rebuild_source\(\s*&mut self,\s*database: &Database,\s*source_id: i64,\s*model_id: u32,\s*model_version: u32,\s*\) -> Result<\(\), eyre::Report>",
                r"pub fn search_vector\(&self, sources: &\[i64\], num_results: usize, vector: Vec<f32>\) -> Vec<SearchItem>",
                r"pub fn search\(&self, model: &Model, sources: &\[i64\], num_results: usize, query: &str\) -> Vec<SearchItem>",
                r"pub fn search_vector_and_retrieve\(\s*&self,\s*database: &Database,\s*sources: &\[i64\],\s*num_results: usize,\s*vector: Vec<f32>,\s*\) -> Result<Vec<\(Item, SearchItem\)>, DbError>",
                r"pub fn search_and_retrieve\(\s*&self,\s*database: &Database,\s*model: &Model,\s*sources: &\[i64\],\s*num_results: usize,\s*query: &str,\s*\) -> Result<Vec<\(Item, SearchItem\)>, DbError>",
                r"pub fn encode_query\(model: &Model, query: &str\) -> Vec<f32>",
                r"pub fn deserialize_embedding\(value: &\[u8\]\) -> Vec<f32>",
                r"pub fn serialize_embedding\(embedding: &\[f32\]\) -> Vec<u8>"):
        assert re.search(pat, search), pat
    variants = re.search(r"pub enum SentenceEmbeddingsModelType \{(.*?)\}", configs, flags=re.S).group(1)
    assert [v.strip() for v in variants.split(",") if v.strip()] == [
        "AllMiniLmL6V2", "AllMiniLmL12V2", "DistiluseBaseMultilingualCased", "AllDistilrobertaV1", "ParaphraseAlbertSmallV2",
        "MsMarcoDistilbertDotV5", "MsMarcoDistilbertBaseTasB", "MsMarcoBertBaseDotV5"]  # configs.rs:30-39; the order is model_id()
    assert "pub fn model_id(&self) -> u32" in configs
    # the directory names the shim asks the library for are the reference's (configs.rs:121-141 for the local ones)
    from perceive_amd import _ffi

    names = [_ffi.lib().pcv_model_type_dir_name(i).decode() for i in range(8)]
    assert names[5:] == ["msmarco-distilbert-dot-v5", "msmarco-distilbert-base-tas-b", "msmarco-bert-base-dot-v5"]
    assert names[0] == "all-MiniLM-L6-v2" and _ffi.lib().pcv_model_type_dir_name(8) is None
    for f in ("README.md", "build.rs", "src/ffi.rs", "src/hip.rs", "src/model.rs", "src/search.rs", "src/configs.rs"):
        text = open(os.path.join(SHIM, f)).read()
        assert "NOT COMPILED" in text or "NOT compiled" in text, f  # says plainly that no toolchain built it
