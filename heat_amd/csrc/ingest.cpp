// ingest.cpp — LightGCN text ingest ("user item item ...\n" per line), the step before the hot path
// (SURVEY §8f row 4).  Replaces the per-line Python loop of /root/reference/cf_cpu/cf/datasets.py:31-79 for the
// interaction list: one pass over an mmap'ed file, manual integer parsing, output in FILE ORDER exactly as
// datasets.py:74-78 appends (user, item) pairs.  Host-only code, no GPU involved.
#include "../../include/heat_cf.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" int heat_cf_set_error_(int code, const char* msg); // engine.cpp

extern "C" int heat_cf_parse_lightgcn(const char* path, char separator, heat_cf_lightgcn* out)
{
    if (!path || !out) return heat_cf_set_error_(HEAT_CF_EINVAL, "path / out is NULL");
    std::memset(out, 0, sizeof(*out));
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return heat_cf_set_error_(HEAT_CF_EINVAL, (std::string("cannot open ") + path).c_str());
    struct stat st;
    if (::fstat(fd, &st) != 0) { ::close(fd); return heat_cf_set_error_(HEAT_CF_EINVAL, "fstat failed"); }
    const size_t size = (size_t)st.st_size;
    const char* data = nullptr;
    if (size)
    {
        data = static_cast<const char*>(::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0));
        if (data == MAP_FAILED) { ::close(fd); return heat_cf_set_error_(HEAT_CF_ENOMEM, "mmap failed"); }
    }
    std::vector<uint64_t> clicks, line_user, line_start;
    try
    {
        clicks.reserve(size / 3);
        const char* p = data;
        const char* end = data + size;
        uint64_t max_user = 0, max_item = 0;
        bool any_user = false, any_item = false;
        while (p < end)
        {
            // one line: strip(), split(separator) (datasets.py:48-51); empty tokens are skipped
            const char* eol = static_cast<const char*>(std::memchr(p, '\n', (size_t)(end - p)));
            if (!eol) eol = end;
            const char* q = p;
            bool have_user = false;
            uint64_t user = 0;
            while (q < eol)
            {
                while (q < eol && (*q == separator || *q == ' ' || *q == '\t' || *q == '\r')) ++q;
                if (q >= eol) break;
                uint64_t v = 0;
                const char* t = q;
                while (q < eol && *q >= '0' && *q <= '9') v = v * 10 + (uint64_t)(*q++ - '0');
                if (q == t || (q < eol && *q != separator && *q != ' ' && *q != '\t' && *q != '\r'))
                {
                    ::munmap(const_cast<char*>(data), size);
                    ::close(fd);
                    return heat_cf_set_error_(HEAT_CF_EINVAL, "non-numeric token in LightGCN file");
                }
                if (!have_user)
                {
                    have_user = true;
                    user = v;
                    line_user.push_back(user);
                    line_start.push_back(clicks.size() / 2);
                    if (!any_user || user > max_user) max_user = user;
                    any_user = true;
                }
                else
                {
                    clicks.push_back(user);
                    clicks.push_back(v);
                    if (!any_item || v > max_item) max_item = v;
                    any_item = true;
                }
            }
            p = eol < end ? eol + 1 : end;
        }
        line_start.push_back(clicks.size() / 2);
        out->num_lines = line_user.size();
        out->n_interactions = clicks.size() / 2;
        out->max_user_id = any_user ? max_user : 0;
        out->max_item_id = any_item ? max_item : 0;
        out->clicks = static_cast<uint64_t*>(std::malloc(std::max<size_t>(clicks.size(), 1) * sizeof(uint64_t)));
        out->line_user = static_cast<uint64_t*>(std::malloc(std::max<size_t>(line_user.size(), 1) * sizeof(uint64_t)));
        out->line_start = static_cast<uint64_t*>(std::malloc(line_start.size() * sizeof(uint64_t)));
        if (!out->clicks || !out->line_user || !out->line_start) throw std::bad_alloc();
        if (!clicks.empty()) std::memcpy(out->clicks, clicks.data(), clicks.size() * sizeof(uint64_t));
        if (!line_user.empty()) std::memcpy(out->line_user, line_user.data(), line_user.size() * sizeof(uint64_t));
        std::memcpy(out->line_start, line_start.data(), line_start.size() * sizeof(uint64_t));
    }
    catch (const std::bad_alloc&)
    {
        heat_cf_free_lightgcn(out);
        if (size) ::munmap(const_cast<char*>(data), size);
        ::close(fd);
        return heat_cf_set_error_(HEAT_CF_ENOMEM, "host allocation failed");
    }
    if (size) ::munmap(const_cast<char*>(data), size);
    ::close(fd);
    return HEAT_CF_OK;
}

extern "C" void heat_cf_free_lightgcn(heat_cf_lightgcn* g)
{
    if (!g) return;
    std::free(g->clicks);
    std::free(g->line_user);
    std::free(g->line_start);
    std::memset(g, 0, sizeof(*g));
}
