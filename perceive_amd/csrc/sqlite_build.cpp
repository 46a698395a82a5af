// sqlite_build.cpp — Searcher::build / rebuild_source straight from the reference's SQLite file
// (crates/perceive-core/search.rs:38-155): the two read queries of build_sources run here and their rows
// stream into the device segments through the public ingestion entry points (pcv_searcher_reserve /
// add_blobs / finalize), so a C or C++ host needs no SQLite code of its own.  Storage itself (schema,
// migrations, writes: db.rs) stays the host's business.  SQLite is bound at run time (libsqlite3.so.0 is on
// every system that has Python; its header is not needed: the handful of prototypes used are declared here).
#include <dlfcn.h>

#include <mutex>

#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.h"

using namespace pcv;

namespace {

struct Sqlite {
    int (*open_v2)(const char*, void**, int, const char*) = nullptr;
    int (*close)(void*) = nullptr;
    int (*prepare_v2)(void*, const char*, int, void**, const char**) = nullptr;
    int (*bind_int64)(void*, int, long long) = nullptr;
    int (*step)(void*) = nullptr;
    long long (*column_int64)(void*, int) = nullptr;
    const void* (*column_blob)(void*, int) = nullptr;
    int (*column_bytes)(void*, int) = nullptr;
    int (*column_type)(void*, int) = nullptr;
    int (*finalize)(void*) = nullptr;
    const char* (*errmsg)(void*) = nullptr;
    bool ok = false;
};

Sqlite& sqlite() {
    static Sqlite q;
    static std::once_flag once;  // first use may come from two searchers on two threads
    std::call_once(once, [] {
        void* h = nullptr;
        for (const char* name : {"libsqlite3.so.0", "libsqlite3.so"})
            if (!h) h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) {
#define PCV_SQL(field, sym) q.field = (decltype(q.field))dlsym(h, sym)
            PCV_SQL(open_v2, "sqlite3_open_v2");
            PCV_SQL(close, "sqlite3_close");
            PCV_SQL(prepare_v2, "sqlite3_prepare_v2");
            PCV_SQL(bind_int64, "sqlite3_bind_int64");
            PCV_SQL(step, "sqlite3_step");
            PCV_SQL(column_int64, "sqlite3_column_int64");
            PCV_SQL(column_blob, "sqlite3_column_blob");
            PCV_SQL(column_bytes, "sqlite3_column_bytes");
            PCV_SQL(column_type, "sqlite3_column_type");
            PCV_SQL(finalize, "sqlite3_finalize");
            PCV_SQL(errmsg, "sqlite3_errmsg");
#undef PCV_SQL
            q.ok = q.open_v2 && q.close && q.prepare_v2 && q.bind_int64 && q.step && q.column_int64 && q.column_blob &&
                   q.column_bytes && q.column_type && q.finalize && q.errmsg;
        }
    });
    return q;
}

constexpr int kSqliteRow = 100, kSqliteDone = 101, kSqliteOpenReadonly = 1, kSqliteBlob = 4;

struct Db {
    void* h = nullptr;
    ~Db() {
        if (h) sqlite().close(h);
    }
};
struct Stmt {
    void* h = nullptr;
    ~Stmt() {
        if (h) sqlite().finalize(h);
    }
};

void prepare(Db& db, Stmt& st, const char* sql) {
    if (sqlite().prepare_v2(db.h, sql, -1, &st.h, nullptr) != 0)
        PCV_FAIL(PCV_ERR_IO, "sqlite: %s (%s)", sqlite().errmsg(db.h), sql);
}

void check(pcv_status st) {
    if (st != PCV_OK) throw Error{st};
}

}  // namespace

extern "C" {

pcv_status pcv_searcher_load_sqlite(pcv_searcher* s, const char* db_path, uint32_t model_id, uint32_t model_version,
                                    const int64_t* only_source, int64_t* out_rows) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr && db_path != nullptr, "searcher_load_sqlite: NULL argument");
        if (!sqlite().ok) PCV_FAIL(PCV_ERR_UNSUPPORTED, "searcher_load_sqlite: libsqlite3.so.0 is not installed");
        int dim = 0;
        check(pcv_searcher_dim(s, &dim));
        Db db;
        if (sqlite().open_v2(db_path, &db.h, kSqliteOpenReadonly, nullptr) != 0)
            PCV_FAIL(PCV_ERR_IO, "sqlite: cannot open %s: %s", db_path, db.h ? sqlite().errmsg(db.h) : "out of memory");
        // the sources to (re)build: all of them (Searcher::build, search.rs:45-48) or one (rebuild_source)
        // One source: its new rows are staged under PCV_STAGING_SOURCE and take the old rows' place only once every one
        // of them has been read, checked and packed (search.rs:57-79 builds the new SourceSearch before it swaps).
        std::vector<int64_t> sources;
        if (only_source) {
            PCV_REQUIRE(*only_source != PCV_STAGING_SOURCE, "searcher_load_sqlite: source id %lld is reserved", (long long)*only_source);
            sources.push_back(*only_source);
            check(pcv_searcher_clear_source(s, PCV_STAGING_SOURCE));  // (left over from a failed call, if anything)
        } else {
            Stmt st;
            prepare(db, st, "SELECT id FROM sources");
            int rc;
            while ((rc = sqlite().step(st.h)) == kSqliteRow) sources.push_back((int64_t)sqlite().column_int64(st.h, 0));
            if (rc != kSqliteDone) PCV_FAIL(PCV_ERR_IO, "sqlite: %s", sqlite().errmsg(db.h));
        }
        std::map<int64_t, size_t> index;
        for (size_t i = 0; i < sources.size(); ++i) index[sources[i]] = i;
        // rows per source first (search.rs:115-140 sizes each source's index from its count): one segment each
        {
            Stmt st;
            prepare(db, st,
                    "SELECT source_id, COUNT(*) FROM items JOIN item_embeddings ie ON model_id=? AND model_version=? AND "
                    "ie.item_id=items.id WHERE skipped IS NULL AND hidden_at IS NULL GROUP BY source_id");
            sqlite().bind_int64(st.h, 1, (long long)model_id);
            sqlite().bind_int64(st.h, 2, (long long)model_version);
            int rc;
            while ((rc = sqlite().step(st.h)) == kSqliteRow) {
                const int64_t src = (int64_t)sqlite().column_int64(st.h, 0);
                if (index.count(src)) check(pcv_searcher_reserve(s, only_source ? PCV_STAGING_SOURCE : src, (int64_t)sqlite().column_int64(st.h, 1)));
            }
            if (rc != kSqliteDone) PCV_FAIL(PCV_ERR_IO, "sqlite: %s", sqlite().errmsg(db.h));
        }
        // the join of search.rs:87-93, streamed: blobs go to the device in chunks per source
        constexpr size_t kChunkRows = 8192;
        struct Pending {
            std::vector<int64_t> ids;
            std::vector<uint8_t> blobs;
        };
        std::vector<Pending> pend(sources.size());
        const size_t row_bytes = (size_t)dim * 4;
        int64_t total = 0;
        auto flush = [&](size_t i) {
            Pending& p = pend[i];
            if (p.ids.empty()) return;
            check(pcv_searcher_add_blobs(s, only_source ? PCV_STAGING_SOURCE : sources[i], p.ids.data(), p.blobs.data(), (int64_t)p.ids.size()));
            total += (int64_t)p.ids.size();
            p.ids.clear();
            p.blobs.clear();
        };
        try {
            Stmt st;
            prepare(db, st,
                    "SELECT items.id, source_id, embedding FROM items JOIN item_embeddings ie ON model_id=? AND model_version=? "
                    "AND ie.item_id=items.id WHERE skipped IS NULL AND hidden_at IS NULL");
            sqlite().bind_int64(st.h, 1, (long long)model_id);
            sqlite().bind_int64(st.h, 2, (long long)model_version);
            int rc;
            while ((rc = sqlite().step(st.h)) == kSqliteRow) {
                const int64_t src = (int64_t)sqlite().column_int64(st.h, 1);
                auto it = index.find(src);
                if (it == index.end()) continue;  // search.rs:106-109: rows of other sources are skipped
                const int64_t id = (int64_t)sqlite().column_int64(st.h, 0);
                if (sqlite().column_type(st.h, 2) != kSqliteBlob)
                    PCV_FAIL(PCV_ERR_IO, "sqlite: embedding of item %lld is not a blob", (long long)id);
                const void* blob = sqlite().column_blob(st.h, 2);
                const size_t nb = (size_t)sqlite().column_bytes(st.h, 2);
                if (nb != row_bytes)
                    PCV_FAIL(PCV_ERR_INVALID, "embedding of item %lld has %zu bytes, the index is %d-d", (long long)id, nb, dim);
                Pending& p = pend[it->second];
                p.ids.push_back(id);
                p.blobs.insert(p.blobs.end(), (const uint8_t*)blob, (const uint8_t*)blob + nb);
                if (p.ids.size() >= kChunkRows) flush(it->second);
            }
            if (rc != kSqliteDone) PCV_FAIL(PCV_ERR_IO, "sqlite: %s", sqlite().errmsg(db.h));
            for (size_t i = 0; i < pend.size(); ++i) flush(i);
            if (only_source) {
                check(pcv_searcher_finalize(s));  // the staged rows are complete (scales, screening copies): now the swap
                check(pcv_searcher_replace_source(s, PCV_STAGING_SOURCE, *only_source));
            }
        } catch (...) {
            if (only_source) {  // the old rows of the source stay what they were: drop the half-built replacement
                const std::string why = pcv::last_error();
                (void)pcv_searcher_clear_source(s, PCV_STAGING_SOURCE);
                (void)pcv_searcher_finalize(s);
                pcv::set_error("%s", why.c_str());
            }
            throw;
        }
        check(pcv_searcher_finalize(s));  // set_searching_mode, search.rs:150-152
        if (out_rows) *out_rows = total;
    });
}

}  // extern "C"
