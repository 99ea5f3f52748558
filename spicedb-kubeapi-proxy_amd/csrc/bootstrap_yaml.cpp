// bootstrap_yaml.cpp -- acl_load_bootstrap_yaml: the bootstrap FILE form of the reference.
//
// The reference boots its embedded engine from YAML documents `{schema: <text>, relationships: <lines>}`: the embedded default
// (pkg/spicedb/bootstrap.yaml:1-40), a file path taken from the endpoint URL (pkg/proxy/options.go:313-316, spicedb.go:22-23) or a byte map
// (spicedb.go:19-21; e2e/embedded_integration_test.go:34-250 builds one per test).  SpiceDB reads them with a YAML library; the two keys the
// proxy's files use are top-level scalars, almost always block scalars, so this reader covers exactly that subset of YAML and refuses the
// rest loudly instead of guessing:
//   * documents separated by `---` lines (several files: the shim joins them); schemas and relationship lists are concatenated;
//   * top-level keys at column 0; `schema` and `relationships` are read, every other key (`schemaFile`, `assertions`, `validation`, ...)
//     is skipped together with its indented body;
//   * values: block scalars `|` / `>` with chomping `-` / `+` and an optional indentation digit, plain one-line scalars, single- and
//     double-quoted one-line scalars (with the usual escapes), or nothing;
//   * `#` comments and blank lines between keys.
// Anchors, flow collections, multi-line plain scalars, tabs as indentation: INVALID_ARGUMENT.
#include <string>
#include <vector>

#include "engine_internal.hpp"

namespace {

struct Doc {
    std::string schema, relationships;
};

bool is_blank(const std::string &l) { return l.find_first_not_of(" \t\r") == std::string::npos; }
size_t indent_of(const std::string &l) { return l.find_first_not_of(' '); }

// one-line scalar after `key:` -- plain, 'single' or "double" quoted; trailing ` # comment` of a plain scalar is dropped
bool inline_scalar(std::string v, std::string *out, std::string *err) {
    const size_t b = v.find_first_not_of(" \t");
    if (b == std::string::npos) {
        out->clear();
        return true;
    }
    v = v.substr(b);
    while (!v.empty() && (v.back() == ' ' || v.back() == '\t' || v.back() == '\r')) v.pop_back();
    if (v[0] == '\'') {
        std::string o;
        size_t i = 1;
        for (; i < v.size(); i++) {
            if (v[i] == '\'') {
                if (i + 1 < v.size() && v[i + 1] == '\'') {
                    o += '\'';
                    i++;
                } else break;
            } else o += v[i];
        }
        if (i >= v.size()) return *err = "unterminated single-quoted scalar", false;
        *out = o;
        return true;
    }
    if (v[0] == '"') {
        std::string o;
        size_t i = 1;
        for (; i < v.size() && v[i] != '"'; i++) {
            if (v[i] != '\\') {
                o += v[i];
                continue;
            }
            if (++i >= v.size()) break;
            switch (v[i]) {
                case 'n': o += '\n'; break;
                case 't': o += '\t'; break;
                case 'r': o += '\r'; break;
                case '0': o += '\0'; break;
                case '"': case '\\': case '/': o += v[i]; break;
                default: return *err = "unsupported escape in double-quoted scalar", false;
            }
        }
        if (i >= v.size()) return *err = "unterminated double-quoted scalar", false;
        *out = o;
        return true;
    }
    if (v[0] == '&' || v[0] == '*' || v[0] == '[' || v[0] == '{' || v[0] == '!') return *err = "anchors, tags and flow collections are not supported in a bootstrap file", false;
    const size_t c = v.find(" #");
    if (c != std::string::npos) v = v.substr(0, c);
    while (!v.empty() && (v.back() == ' ' || v.back() == '\t')) v.pop_back();
    *out = v;
    return true;
}

bool parse(const std::string &text, std::vector<Doc> *docs, std::string *err) {
    std::vector<std::string> lines;
    for (size_t p = 0; p <= text.size();) {
        size_t e = text.find('\n', p);
        if (e == std::string::npos) e = text.size();
        std::string l = text.substr(p, e - p);
        if (!l.empty() && l.back() == '\r') l.pop_back();
        lines.push_back(std::move(l));
        p = e + 1;
    }
    docs->emplace_back();
    for (size_t i = 0; i < lines.size();) {
        const std::string &l = lines[i];
        if (is_blank(l) || l[0] == '#') {
            i++;
            continue;
        }
        if (l.compare(0, 3, "---") == 0 && (l.size() == 3 || l[3] == ' ')) {
            docs->emplace_back();
            i++;
            continue;
        }
        if (l.compare(0, 3, "...") == 0) {
            i++;
            continue;
        }
        if (l[0] == ' ' || l[0] == '\t') return *err = "line " + std::to_string(i + 1) + ": unexpected indentation at the top level", false;
        const size_t colon = l.find(':');
        if (colon == std::string::npos || (colon + 1 < l.size() && l[colon + 1] != ' ' && l[colon + 1] != '\t'))
            return *err = "line " + std::to_string(i + 1) + ": expected `key: value`", false;
        std::string key = l.substr(0, colon);
        if (key.size() >= 2 && (key.front() == '"' || key.front() == '\'') && key.back() == key.front()) key = key.substr(1, key.size() - 2);
        std::string rest = l.substr(colon + 1);
        const size_t vb = rest.find_first_not_of(" \t");
        std::string value;
        i++;
        if (vb != std::string::npos && (rest[vb] == '|' || rest[vb] == '>')) {
            // block scalar header: | or >, then chomping and / or an indentation digit in either order, then an optional comment
            const bool folded = rest[vb] == '>';
            char chomp = 'c';
            size_t want_indent = 0;
            size_t q = vb + 1;
            for (; q < rest.size() && rest[q] != ' ' && rest[q] != '\t'; q++) {
                if (rest[q] == '-' || rest[q] == '+') chomp = rest[q];
                else if (rest[q] >= '1' && rest[q] <= '9') want_indent = (size_t)(rest[q] - '0');
                else return *err = "line " + std::to_string(i) + ": bad block scalar header", false;
            }
            const size_t tail = rest.find_first_not_of(" \t", q);
            if (tail != std::string::npos && rest[tail] != '#') return *err = "line " + std::to_string(i) + ": text after a block scalar header", false;
            size_t ind = want_indent;
            std::vector<std::string> body;
            for (; i < lines.size(); i++) {
                const std::string &b = lines[i];
                if (is_blank(b)) {
                    body.emplace_back();
                    continue;
                }
                if (b[0] == '\t') return *err = "line " + std::to_string(i + 1) + ": tab used as indentation", false;
                const size_t bi = indent_of(b);
                if (!ind) ind = bi;
                if (bi < ind || ind == 0) break;
                body.push_back(b.substr(ind));
            }
            while (!body.empty() && body.back().empty() && chomp != '+') body.pop_back();  // (trailing blank lines belong to the chomping)
            if (folded) {
                for (size_t k = 0; k < body.size(); k++) {
                    if (body[k].empty()) value += '\n';
                    else {
                        if (!value.empty() && value.back() != '\n') value += ' ';
                        value += body[k];
                    }
                }
            } else {
                for (size_t k = 0; k < body.size(); k++) {
                    if (k) value += '\n';
                    value += body[k];
                }
            }
            if (chomp != '-' && !body.empty()) value += '\n';
        } else {
            if (!inline_scalar(rest, &value, err)) return *err = "line " + std::to_string(i) + ": " + *err, false;
            // the indented body of a key we skip (a mapping or a sequence), or a multi-line plain scalar (refused for the keys we read)
            const size_t first_body = i;
            while (i < lines.size() && (is_blank(lines[i]) || lines[i][0] == ' ' || lines[i][0] == '\t' || (lines[i].compare(0, 2, "- ") == 0))) i++;
            bool had_body = false;
            for (size_t k = first_body; k < i; k++) had_body = had_body || !is_blank(lines[k]);
            if (had_body && (key == "schema" || key == "relationships")) return *err = "key `" + key + "`: only block scalars and one-line scalars are supported", false;
        }
        Doc &d = docs->back();
        if (key == "schema") {
            if (!d.schema.empty()) return *err = "duplicate key `schema` in one document", false;
            d.schema = value;
        } else if (key == "relationships") {
            if (!d.relationships.empty()) return *err = "duplicate key `relationships` in one document", false;
            d.relationships = value;
        }
    }
    return true;
}

}  // namespace

extern "C" int acl_load_bootstrap_yaml(acl_engine_t *h, const char *yaml, size_t len) {
    if (!yaml) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_load_bootstrap_yaml: yaml is NULL");
    std::vector<Doc> docs;
    std::string err;
    if (!parse(std::string(yaml, len), &docs, &err)) return fail(ACL_ERR_INVALID_ARGUMENT, "bootstrap yaml: " + err);
    std::string schema, rels;
    for (const Doc &d : docs) {
        if (!d.schema.empty()) schema += (schema.empty() ? "" : "\n") + d.schema;
        if (!d.relationships.empty()) rels += (rels.empty() || rels.back() == '\n' ? "" : "\n") + d.relationships;
    }
    if (schema.find_first_not_of(" \t\r\n") == std::string::npos) return fail(ACL_ERR_INVALID_ARGUMENT, "bootstrap yaml: no `schema` key");
    return acl_load_bootstrap(h, schema.data(), schema.size(), rels.empty() ? nullptr : rels.data(), rels.size());
}
