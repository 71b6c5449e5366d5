// Minimal XML reader for URDF and COLLADA files: elements, attributes, text, comments,
// processing instructions, self-closing tags. No entities beyond the five predefined ones,
// no DTDs, no namespaces (prefixes are kept as part of the tag name). Header-only.
// The files are caller-named: nesting deeper than kMaxDepth is refused (the reader, the tree's destructor and the
// find helpers recurse per level - a file of 100 000 nested tags used to overflow the stack).
#pragma once
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace xmlmin {

constexpr int kMaxDepth = 256;   // URDF nests 5 levels, COLLADA about 10

struct Node {
  std::string tag;
  std::vector<std::pair<std::string, std::string>> attrs;
  std::vector<std::unique_ptr<Node>> children;
  std::string text;

  const std::string *attr(const char *name) const {
    for (auto &a : attrs)
      if (a.first == name) return &a.second;
    return nullptr;
  }
  std::string attr_or(const char *name, const char *dflt) const {
    const std::string *a = attr(name);
    return a ? *a : std::string(dflt);
  }
  const Node *child(const char *name) const {
    for (auto &c : children)
      if (c->tag == name) return c.get();
    return nullptr;
  }
  std::vector<const Node *> all(const char *name) const {
    std::vector<const Node *> out;
    for (auto &c : children)
      if (c->tag == name) out.push_back(c.get());
    return out;
  }
  // depth-first search for the first descendant with this tag
  const Node *find(const char *name) const {
    for (auto &c : children) {
      if (c->tag == name) return c.get();
      if (const Node *n = c->find(name)) return n;
    }
    return nullptr;
  }
  void find_all(const char *name, std::vector<const Node *> &out) const {
    for (auto &c : children) {
      if (c->tag == name) out.push_back(c.get());
      c->find_all(name, out);
    }
  }
};

class Parser {
 public:
  explicit Parser(const std::string &s) : s_(s) {}

  std::unique_ptr<Node> parse() {
    skip_misc();
    auto root = element(0);
    if (!root) fail("no root element");
    return root;
  }

 private:
  const std::string &s_;
  size_t p_ = 0;

  [[noreturn]] void fail(const char *what) const {
    throw std::runtime_error(std::string("XML: ") + what + " at byte " + std::to_string(p_));
  }
  bool starts(const char *lit) const { return s_.compare(p_, std::strlen(lit), lit) == 0; }
  void skip_ws() {
    while (p_ < s_.size() && std::strchr(" \t\r\n", s_[p_])) p_++;
  }
  void skip_until(const char *lit) {
    size_t e = s_.find(lit, p_);
    if (e == std::string::npos) fail("unterminated construct");
    p_ = e + std::strlen(lit);
  }
  void skip_misc() {
    for (;;) {
      skip_ws();
      if (starts("<?")) skip_until("?>");
      else if (starts("<!--")) skip_until("-->");
      else if (starts("<!")) skip_until(">");
      else return;
    }
  }
  static std::string unescape(const std::string &in) {
    if (in.find('&') == std::string::npos) return in;
    std::string out;
    for (size_t i = 0; i < in.size(); i++) {
      if (in[i] != '&') { out += in[i]; continue; }
      static const char *ent[][2] = {{"&lt;", "<"}, {"&gt;", ">"}, {"&amp;", "&"}, {"&quot;", "\""}, {"&apos;", "'"}};
      bool hit = false;
      for (auto &e : ent)
        if (in.compare(i, std::strlen(e[0]), e[0]) == 0) { out += e[1]; i += std::strlen(e[0]) - 1; hit = true; break; }
      if (!hit) out += in[i];
    }
    return out;
  }
  std::string name() {
    size_t b = p_;
    while (p_ < s_.size() && !std::strchr(" \t\r\n/>=", s_[p_])) p_++;
    if (p_ == b) fail("expected a name");
    return s_.substr(b, p_ - b);
  }
  std::unique_ptr<Node> element(int depth) {
    if (p_ >= s_.size() || s_[p_] != '<') return nullptr;
    if (depth > kMaxDepth) fail("elements nested too deeply");
    p_++;
    auto n = std::make_unique<Node>();
    n->tag = name();
    for (;;) {
      skip_ws();
      if (p_ >= s_.size()) fail("unterminated start tag");
      if (s_[p_] == '/') {
        if (!starts("/>")) fail("bad empty-element tag");
        p_ += 2;
        return n;
      }
      if (s_[p_] == '>') { p_++; break; }
      std::string k = name();
      skip_ws();
      if (p_ >= s_.size() || s_[p_] != '=') fail("attribute without value");
      p_++;
      skip_ws();
      char q = s_[p_];
      if (q != '"' && q != '\'') fail("unquoted attribute value");
      size_t e = s_.find(q, p_ + 1);
      if (e == std::string::npos) fail("unterminated attribute value");
      n->attrs.emplace_back(k, unescape(s_.substr(p_ + 1, e - p_ - 1)));
      p_ = e + 1;
    }
    for (;;) {
      size_t lt = s_.find('<', p_);
      if (lt == std::string::npos) fail("unterminated element");
      n->text += s_.substr(p_, lt - p_);
      p_ = lt;
      if (starts("<!--")) { skip_until("-->"); continue; }
      if (starts("<![CDATA[")) {
        size_t e = s_.find("]]>", p_);
        if (e == std::string::npos) fail("unterminated CDATA");
        n->text += s_.substr(p_ + 9, e - p_ - 9);
        p_ = e + 3;
        continue;
      }
      if (starts("<?")) { skip_until("?>"); continue; }
      if (starts("</")) {
        p_ += 2;
        std::string close = name();
        if (close != n->tag) fail("mismatched end tag");
        skip_ws();
        if (p_ >= s_.size() || s_[p_] != '>') fail("bad end tag");
        p_++;
        n->text = unescape(n->text);
        return n;
      }
      n->children.push_back(element(depth + 1));
    }
  }
};

inline std::unique_ptr<Node> parse(const std::string &text) { return Parser(text).parse(); }

}  // namespace xmlmin
