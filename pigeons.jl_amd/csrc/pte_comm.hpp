// pte_comm.hpp -- RCCL reached at run time (dlopen), so that libpte.so has no link-time dependency on
// librccl and shares whichever copy the host process already mapped (PyTorch-ROCm bundles its own
// librccl.so with the same SONAME; a Julia host gets /opt/rocm/lib/librccl.so.1).
//
// Only the point-to-point pair that the boundary exchange needs (ncclSend / ncclRecv inside one group on
// the engine's stream) plus the three small collectives a host without MPI wants around a round
// (barrier = all-reduce of one word, all-reduce MAX for timing, all-gather of the reduced recorders).
// This is what replaces the reference's MPI transport, src/mpi_utils/Entangler.jl:118-180 (`transmit!`)
// and :188-251 (`all_reduce_deterministically` -- here recorders are keyed by chain / pair, so a plain
// gather in rank order is already deterministic).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <string>

namespace pte {

struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string where;
};

// $PTE_RCCL_LIB names another library to take the 12 entry points from (tests hand in a stand-in that accepts two ranks on one
// device, tests/fakerccl).  It is honoured ONLY after the host opted in with pte_comm_allow_library_override(1): a stale variable in
// a production environment must not silently re-route the boundary traffic, so without the opt-in a set variable is an ERROR.
inline int &rccl_override_allowed() { static int allowed = 0; return allowed; }

// Returns nullptr and fills `err` when no RCCL can be mapped.  Search order: $PTE_RCCL_LIB (explicit override, opt-in only), a copy
// already in the process (RTLD_NOLOAD by SONAME), the loader's search path, /opt/rocm/lib.
inline RcclApi *rccl_api(std::string &err) {
    static RcclApi api;
    static bool tried = false;
    static std::string first_err;
    if (api.lib) return &api;
    if (tried) { err = first_err; return nullptr; }
    tried = true;
    // an explicit $PTE_RCCL_LIB wins over everything (also over a copy the process has already mapped: tests hand in a
    // stand-in this way, tests/fakerccl); an unloadable override is an error, not a reason to fall through to another library
    void *lib = nullptr;
    std::string where;
    const char *env = std::getenv("PTE_RCCL_LIB");
    if (env && *env && !rccl_override_allowed()) {
        tried = false;                                   // (not final: the host may opt in and call again)
        err = std::string("$PTE_RCCL_LIB=") + env + " is set, but the host did not opt in to a transport override "
              "(pte_comm_allow_library_override(1)): refusing to route the boundary exchange through it -- unset the variable to use RCCL";
        return nullptr;
    }
    if (env && *env) {
        lib = dlopen(env, RTLD_NOW | RTLD_LOCAL);
        where = env;
        if (!lib) {
            const char *e = dlerror();
            first_err = std::string("RCCL override $PTE_RCCL_LIB=") + env + " cannot be loaded (" + (e ? e : "?") + ")";
            err = first_err;
            return nullptr;
        }
    }
    if (!lib) { lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD); where = "librccl.so.1 (already mapped)"; }
    if (!lib) {
        const char *cands[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char *c : cands) {
            lib = dlopen(c, RTLD_NOW | RTLD_LOCAL);
            if (lib) { where = c; break; }
        }
    }
    if (!lib) {
        const char *e = dlerror();
        first_err = std::string("RCCL is not available (dlopen librccl.so.1: ") + (e ? e : "not found") + ")";
        err = first_err;
        return nullptr;
    }
    bool ok = true;
    auto sym = [&](const char *name) { void *p = dlsym(lib, name); if (!p) { ok = false; first_err = std::string("RCCL lacks symbol ") + name; } return p; };
#define PTE_RCCL_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(sym(name))
    PTE_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    PTE_RCCL_SYM(CommInitRank, "ncclCommInitRank");
    PTE_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    PTE_RCCL_SYM(CommCount, "ncclCommCount");
    PTE_RCCL_SYM(Send, "ncclSend");
    PTE_RCCL_SYM(Recv, "ncclRecv");
    PTE_RCCL_SYM(GroupStart, "ncclGroupStart");
    PTE_RCCL_SYM(GroupEnd, "ncclGroupEnd");
    PTE_RCCL_SYM(AllReduce, "ncclAllReduce");
    PTE_RCCL_SYM(AllGather, "ncclAllGather");
    PTE_RCCL_SYM(GetErrorString, "ncclGetErrorString");
    PTE_RCCL_SYM(GetVersion, "ncclGetVersion");
#undef PTE_RCCL_SYM
    if (!ok) { err = first_err; return nullptr; }
    api.lib = lib;
    api.where = where;
    {   // the file the entry points really come from (a SONAME or "already mapped" says little)
        Dl_info di;
        if (dladdr(reinterpret_cast<void *>(api.CommInitRank), &di) && di.dli_fname) api.where = std::string(di.dli_fname) + (env && *env ? " (override $PTE_RCCL_LIB)" : "");
    }
    if (env && *env) {
        int v = 0; api.GetVersion(&v);
        std::fprintf(stderr, "[pte] transport OVERRIDE: ncclSend / ncclRecv taken from %s (reports version %d) instead of the process's RCCL\n", api.where.c_str(), v);
    }
    return &api;
}

}  // namespace pte
