"""tools/check_triton_header.py — step 0 of a real deployment: include/tritonbackend_hps.h restates the part of Triton's C API the
plugin imports (the real headers are fetched over the network by the reference's build, hps_backend/CMakeLists.txt:82-100, and do
not exist in this image); the checker diffs it against the real tritonbackend.h / tritonserver.h of the target release.  Here the
checker itself is tested, against small synthetic headers written the way the real ones are (DECLSPEC macros, `struct X*`
parameters, enums without explicit values, one declaration over several lines, comments)."""
import importlib.util
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
spec = importlib.util.spec_from_file_location("check_triton_header", ROOT / "tools" / "check_triton_header.py")
cth = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cth)

OURS = (ROOT / "include" / "tritonbackend_hps.h").read_text()


def _real_style(ours: str):
    """The two real headers, synthesised from our restatement in the real headers' style: every function declared as
    `TRITON*_DECLSPEC struct TRITONSERVER_Error* Name(\n    struct X* x, ...);`, enums without explicit values, flags as enums."""
    funcs = cth.parse_functions(ours)
    server, backend = ["#pragma once\n#include <stdint.h>\n/// \\file\n"], ["#pragma once\n#include \"tritonserver.h\"\n"]
    backend.append("#define TRITONBACKEND_API_VERSION_MAJOR 1\n#define TRITONBACKEND_API_VERSION_MINOR 19\n")
    server.append("""
typedef enum TRITONSERVER_datatype_enum {
  TRITONSERVER_TYPE_INVALID,
  TRITONSERVER_TYPE_BOOL,
  TRITONSERVER_TYPE_UINT8,
  TRITONSERVER_TYPE_UINT16,
  TRITONSERVER_TYPE_UINT32,
  TRITONSERVER_TYPE_UINT64,
  TRITONSERVER_TYPE_INT8,
  TRITONSERVER_TYPE_INT16,
  TRITONSERVER_TYPE_INT32,
  TRITONSERVER_TYPE_INT64,
  TRITONSERVER_TYPE_FP16,
  TRITONSERVER_TYPE_FP32,
  TRITONSERVER_TYPE_FP64,
  TRITONSERVER_TYPE_BYTES,
  TRITONSERVER_TYPE_BF16
} TRITONSERVER_DataType;
/// memory types
typedef enum TRITONSERVER_memorytype_enum { TRITONSERVER_MEMORY_CPU, TRITONSERVER_MEMORY_CPU_PINNED, TRITONSERVER_MEMORY_GPU } TRITONSERVER_MemoryType;
typedef enum TRITONSERVER_errorcode_enum {
  TRITONSERVER_ERROR_UNKNOWN, TRITONSERVER_ERROR_INTERNAL, TRITONSERVER_ERROR_NOT_FOUND, TRITONSERVER_ERROR_INVALID_ARG,
  TRITONSERVER_ERROR_UNAVAILABLE, TRITONSERVER_ERROR_UNSUPPORTED, TRITONSERVER_ERROR_ALREADY_EXISTS, TRITONSERVER_ERROR_CANCELLED
} TRITONSERVER_Error_Code;
typedef enum TRITONSERVER_loglevel_enum { TRITONSERVER_LOG_INFO, TRITONSERVER_LOG_WARN, TRITONSERVER_LOG_ERROR, TRITONSERVER_LOG_VERBOSE } TRITONSERVER_LogLevel;
typedef enum TRITONSERVER_instancegroupkind_enum {
  TRITONSERVER_INSTANCEGROUPKIND_AUTO, TRITONSERVER_INSTANCEGROUPKIND_CPU, TRITONSERVER_INSTANCEGROUPKIND_GPU, TRITONSERVER_INSTANCEGROUPKIND_MODEL
} TRITONSERVER_InstanceGroupKind;
typedef enum tritonserver_responsecompleteflag_enum { TRITONSERVER_RESPONSE_COMPLETE_FINAL = 1 } TRITONSERVER_ResponseCompleteFlag;
typedef enum tritonserver_requestreleaseflag_enum { TRITONSERVER_REQUEST_RELEASE_ALL = 1, TRITONSERVER_REQUEST_RELEASE_RESCHEDULE = 2 } TRITONSERVER_RequestReleaseFlag;
""")
    backend.append("typedef enum TRITONBACKEND_artifacttype_enum { TRITONBACKEND_ARTIFACT_FILESYSTEM } TRITONBACKEND_ArtifactType;\n")
    for name, (ret, params) in funcs.items():
        decl = "TRITONSERVER_DECLSPEC" if name.startswith("TRITONSERVER_") else ("TRITONBACKEND_ISPEC" if name in (
            "TRITONBACKEND_Initialize", "TRITONBACKEND_Finalize", "TRITONBACKEND_ModelInitialize", "TRITONBACKEND_ModelFinalize",
            "TRITONBACKEND_ModelInstanceInitialize", "TRITONBACKEND_ModelInstanceFinalize", "TRITONBACKEND_ModelInstanceExecute") else "TRITONBACKEND_DECLSPEC")
        def styl(t):
            return t.replace("TRITONSERVER_Error*", "struct TRITONSERVER_Error*").replace("TRITONBACKEND_Request*", "struct TRITONBACKEND_Request*")
        ps = ",\n    ".join(f"{styl(p)} arg{i}" for i, p in enumerate(params)) or "void"
        text = f"/// {name} does what it does (a `;` in a comment; and a (parenthesis)).\n{decl} {styl(ret)} {name}(\n    {ps});\n"
        (server if name.startswith("TRITONSERVER_") else backend).append(text)
    return "\n".join(backend), "\n".join(server)


def test_our_restatement_parses_completely():
    f = cth.parse_functions(OURS)
    assert len(f) >= 52, len(f)          # the 45 imports of SURVEY 8b (+ ErrorCode / ErrorMessage / LogIsEnabled / ResponseDelete) + the 7 exports
    assert f["TRITONBACKEND_ModelInstanceExecute"] == ("TRITONSERVER_Error*", ["TRITONBACKEND_ModelInstance*", "TRITONBACKEND_Request**", "uint32_t"])
    assert f["TRITONBACKEND_InputBuffer"][1] == ["TRITONBACKEND_Input*", "uint32_t", "const void**", "uint64_t*", "TRITONSERVER_MemoryType*", "int64_t*"]
    c = cth.parse_constants(OURS)
    assert (c["TRITONSERVER_TYPE_INT32"], c["TRITONSERVER_TYPE_INT64"], c["TRITONSERVER_TYPE_FP32"]) == (8, 9, 11)   # SURVEY 8b [EXT]
    assert c["TRITONSERVER_MEMORY_GPU"] == 2 and c["TRITONSERVER_ERROR_UNSUPPORTED"] == 5 and c["TRITONSERVER_INSTANCEGROUPKIND_MODEL"] == 3
    assert c["TRITONSERVER_RESPONSE_COMPLETE_FINAL"] == 1 and c["TRITONSERVER_REQUEST_RELEASE_ALL"] == 1
    assert (c["TRITONBACKEND_API_VERSION_MAJOR"], c["TRITONBACKEND_API_VERSION_MINOR"]) == (1, 10)


def test_headers_in_the_real_style_agree_and_every_kind_of_difference_is_found(tmp_path):
    backend, server = _real_style(OURS)
    assert cth.compare(OURS, [backend, server]) == []
    # the command line, as INTEGRATION.md documents it
    (tmp_path / "tritonbackend.h").write_text(backend)
    (tmp_path / "tritonserver.h").write_text(server)
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "check_triton_header.py"), str(tmp_path / "tritonbackend.h"), str(tmp_path / "tritonserver.h")],
                       capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), r.stdout + r.stderr
    # (1) a parameter type changed
    bad = backend.replace("uint64_t* arg3,\n    TRITONSERVER_MemoryType* arg4", "uint32_t* arg3,\n    TRITONSERVER_MemoryType* arg4", 1)
    assert bad != backend
    p = cth.compare(OURS, [bad, server])
    assert len(p) == 1 and "parameter 4" in p[0] and "uint32_t*" in p[0]
    # (2) a parameter added
    bad = backend.replace("TRITONBACKEND_RequestRelease(\n    struct TRITONBACKEND_Request* arg0,", "TRITONBACKEND_RequestRelease(\n    struct TRITONBACKEND_Request* arg0, void* userp,", 1)
    assert bad != backend
    p = cth.compare(OURS, [bad, server])
    assert len(p) == 1 and "TRITONBACKEND_RequestRelease" in p[0] and "parameters" in p[0]
    # (3) an enumerator inserted in the middle (every later value shifts)
    bad = server.replace("TRITONSERVER_TYPE_INT64,", "TRITONSERVER_TYPE_INT48,\n  TRITONSERVER_TYPE_INT64,", 1)
    p = cth.compare(OURS, [backend, bad])
    assert any("TRITONSERVER_TYPE_INT64: 9 here, 10" in x for x in p) and any("TRITONSERVER_TYPE_FP32" in x for x in p)
    # (4) a function the plugin imports is gone
    bad = server.replace("TRITONSERVER_DataTypeString(", "TRITONSERVER_DataTypeToString(", 1)
    p = cth.compare(OURS, [backend, bad])
    assert p == ["TRITONSERVER_DataTypeString: not declared by the real headers"]
    # (5) the API version: a real minor below ours, or another major, is refused (hps.cc:64-82 does the same at load)
    for old, new in (("MINOR 19", "MINOR 9"), ("MAJOR 1", "MAJOR 2")):
        p = cth.compare(OURS, [backend.replace(old, new, 1), server])
        assert len(p) == 1 and "backend API version" in p[0]
    # (6) a return type changed
    bad = server.replace("TRITONSERVER_DECLSPEC const char* TRITONSERVER_ErrorMessage(", "TRITONSERVER_DECLSPEC char* TRITONSERVER_ErrorMessage(", 1)
    assert bad != server
    p = cth.compare(OURS, [backend, bad])
    assert len(p) == 1 and "return type" in p[0]
    (tmp_path / "tritonserver.h").write_text(bad)
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "check_triton_header.py"), str(tmp_path / "tritonbackend.h"), str(tmp_path / "tritonserver.h")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "return type" in r.stdout
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "check_triton_header.py"), str(tmp_path / "nope.h"), str(tmp_path / "tritonserver.h")],
                       capture_output=True, text=True)
    assert r.returncode == 2
