// np_shim_common.h -- what the two reference-side bindings (np_dropin.cpp: the six per-call entry points; np_batch_dropin.cpp:
// the batched callers) share: ONE library context per process and ONE cache of the pore models registered on the device.
// Compiled inside a nanopolish build (C++11), like the files that include it.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>
#include "nanopolish_poremodel.h"
#include "np_hmm.h"

namespace np_shim {

// FNV-1a over the bit patterns of the three per-state doubles the device table holds, every `step`-th state
inline uint64_t model_hash(const PoreModel* m, size_t step)
{
    uint64_t h = 1469598103934665603ull ^ (uint64_t)m->states.size();
    for (size_t i = 0; i < m->states.size(); i += step) {
        const double v[3] = {m->states[i].level_mean, m->states[i].level_stdv, m->states[i].level_log_stdv};
        for (int q = 0; q < 3; ++q) { uint64_t u; memcpy(&u, &v[q], 8); h = (h ^ u) * 1099511628211ull; }
    }
    return h;
}

struct Shim {
    // The cache is keyed by address, and the reference overwrites registered models in place (PoreModelSet::register_model,
    // src/pore_model/nanopolish_pore_model_set.cpp:70 -- methyltrain's add_model every training round): an entry is only valid
    // while the model's content matches what was uploaded.  Per call that is checked with a FINGERPRINT (size + 64 evenly spaced
    // states: a few hundred bytes, so the OpenMP callers of the per-call shim do not serialise on a 100-370 KB hash); the FULL
    // hash runs when the fingerprint differs, after np_dropin_invalidate_models() -- which a caller that edits single states in
    // place (a training round) calls once per round -- and, as a backstop for callers that never call it, on every
    // 256th use of an entry.
    struct Entry { int id; uint64_t fingerprint, hash; size_t n; bool check_full; unsigned uses; };
    // One library context per (process, slot).  Slot 0 is the process-wide default (device NP_DEVICE, or 0): the per-call shim and
    // the synchronous bindings use it.  NpBatchPipeline's multi-device form opens one more slot per listed device -- the same
    // device may be listed twice (two contexts on one GPU: tests) -- and keeps them for the life of the process.
    struct Dev { np_ctx* ctx; int device; bool in_use; std::map<const PoreModel*, Entry> models; Dev() : ctx(NULL), device(0), in_use(false) {} };
    std::vector<Dev*> devs;
    std::mutex lock;
    Shim() {}

    np_ctx* get() { return ctx(0); }

    np_ctx* ctx(int slot)
    {
        std::lock_guard<std::mutex> g(lock);
        if (devs.empty()) devs.push_back(new Dev());
        if (slot < 0 || slot >= (int)devs.size()) { fprintf(stderr, "nanopolish_amd: no context slot %d\n", slot); exit(EXIT_FAILURE); }
        Dev& d = *devs[slot];
        if (!d.ctx) {
            if (slot == 0) { const char* dev = getenv("NP_DEVICE"); d.device = dev ? atoi(dev) : 0; }
            d.ctx = np_create(d.device, NULL);
            if (!d.ctx) { fprintf(stderr, "nanopolish_amd: %s\n", np_last_error(NULL)); exit(EXIT_FAILURE); }
        }
        return d.ctx;
    }

    // a context slot of its own on `device` for the caller (released with release_slot; the context and its model cache stay for the next taker)
    int take_slot(int device)
    {
        int slot = -1;
        {
            std::lock_guard<std::mutex> g(lock);
            if (devs.empty()) devs.push_back(new Dev());
            for (size_t i = 1; i < devs.size() && slot < 0; ++i)
                if (!devs[i]->in_use && devs[i]->device == device) { devs[i]->in_use = true; slot = (int)i; }
            if (slot < 0) {
                Dev* d = new Dev(); d->device = device; d->in_use = true;
                devs.push_back(d);
                slot = (int)devs.size() - 1;          // taken under the lock: two pipelines constructed at once get two slots
            }
        }
        (void)ctx(slot);
        return slot;
    }
    void release_slot(int slot)
    {
        std::lock_guard<std::mutex> g(lock);
        if (slot > 0 && slot < (int)devs.size()) devs[slot]->in_use = false;
    }

    int model_id(const PoreModel* m, int slot = 0)
    {
        np_ctx* c = ctx(slot);
        std::lock_guard<std::mutex> g(lock);
        std::map<const PoreModel*, Entry>& models = devs[slot]->models;
        const size_t n = m->states.size();
        const size_t step = n > 64 ? n / 64 : 1;
        const uint64_t fp = model_hash(m, step);
        std::map<const PoreModel*, Entry>::iterator it = models.find(m);
        if (it != models.end() && it->second.n == n && it->second.fingerprint == fp && !it->second.check_full && (++it->second.uses & 255u) != 0) return it->second.id;
        const uint64_t h = model_hash(m, 1);
        if (it != models.end() && it->second.n == n && it->second.hash == h) {          // invalidated (or the periodic full check), but unchanged
            it->second.fingerprint = fp; it->second.check_full = false;
            return it->second.id;
        }
        std::vector<double> lm(n), ls(n), ll(n);
        for (size_t i = 0; i < n; ++i) { lm[i] = m->states[i].level_mean; ls[i] = m->states[i].level_stdv; ll[i] = m->states[i].level_log_stdv; }
        if (it != models.end() && it->second.n == n) {
            const int rc = np_update_model(c, it->second.id, (int)n, lm.data(), ls.data(), ll.data());
            if (rc != NP_OK) { fprintf(stderr, "nanopolish_amd: np_update_model: %s\n", np_last_error(c)); exit(EXIT_FAILURE); }
            it->second.fingerprint = fp; it->second.hash = h; it->second.check_full = false;
            return it->second.id;
        }
        const int id = np_register_model(c, (int)m->k, (int)n, lm.data(), ls.data(), ll.data());
        if (id < 0) { fprintf(stderr, "nanopolish_amd: np_register_model: %s\n", np_last_error(c)); exit(EXIT_FAILURE); }
        Entry e; e.id = id; e.fingerprint = fp; e.hash = h; e.n = n; e.check_full = false; e.uses = 0;
        models[m] = e;
        return id;
    }

    void invalidate()
    {
        std::lock_guard<std::mutex> g(lock);
        for (size_t d = 0; d < devs.size(); ++d)
            for (std::map<const PoreModel*, Entry>::iterator it = devs[d]->models.begin(); it != devs[d]->models.end(); ++it) it->second.check_full = true;
    }
};

// one instance per process (a function-local static of an inline function is shared by every translation unit of the image)
inline Shim& shim() { static Shim s; return s; }

inline void check(int rc, const char* what)
{
    if (rc != NP_OK) { fprintf(stderr, "nanopolish_amd: %s failed (%d): %s\n", what, rc, np_last_error(shim().get())); exit(EXIT_FAILURE); }
}

inline void die(const char* what)
{
    fprintf(stderr, "nanopolish_amd: %s\n", what);
    exit(EXIT_FAILURE);
}

// arrays laid out back to back in one allocation, each aligned to 256 bytes
struct Layout {
    size_t size;
    Layout() : size(0) {}
    size_t add(size_t bytes) { const size_t o = size; size = (size + bytes + 255) & ~(size_t)255; return o; }
};

// a device allocation, optionally with a pinned host mirror of the same layout; grows, never shrinks
struct Blob {
    char* d; char* h; size_t cap; bool mirrored;
    explicit Blob(bool with_host) : d(NULL), h(NULL), cap(0), mirrored(with_host) {}
    void reserve(np_ctx* c, size_t bytes)
    {
        if (bytes <= cap) return;
        release(c);
        const size_t want = bytes + bytes / 4 + 4096;
        d = (char*)np_dev_alloc(c, want);
        if (!d) die(np_last_error(c));
        if (mirrored) { h = (char*)np_host_alloc(c, want); if (!h) die(np_last_error(c)); }
        cap = want;
    }
    void release(np_ctx* c)
    {
        if (d) np_dev_free(c, d);
        if (h) np_host_free(c, h);
        d = h = NULL; cap = 0;
    }
};

} // namespace np_shim

// For callers that edit registered PoreModels in place one state at a time (methyltrain's rounds): the next call on every cached
// model re-checks its full content.  (An overwrite that touches most states -- PoreModelSet::register_model replacing a model --
// is noticed without it, by the per-call fingerprint.)
extern "C" void np_dropin_invalidate_models(void);
