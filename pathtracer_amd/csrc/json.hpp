// Minimal JSON reader for the reference's conf files.
//
// Accepts what jsoncpp's CharReaderBuilder accepts in the reference's use
// (src/util/cJsonUtil.cpp:6-28): strict JSON plus // and /* */ comments; a duplicated key keeps the
// LAST value (config/cbox_cube.conf repeats "Light_bak").  Missing keys read as null, and the
// as*() accessors on null return 0 / false / "" like jsoncpp's do.
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace ptx
{

class Json
{
public:
    enum Type
    {
        Null,
        Bool,
        Number,
        String,
        Array,
        Object
    };

    Json() : type_(Null), num_(0), b_(false) {}

    Type type() const { return type_; }
    bool isNull() const { return type_ == Null; }
    bool isArray() const { return type_ == Array; }
    bool isObject() const { return type_ == Object; }
    size_t size() const { return type_ == Array ? arr_.size() : (type_ == Object ? obj_.size() : 0); }

    double asDouble() const { return type_ == Number ? num_ : (type_ == Bool ? (b_ ? 1.0 : 0.0) : 0.0); }
    float asFloat() const { return static_cast<float>(asDouble()); }
    int asInt() const { return static_cast<int>(asDouble()); }
    bool asBool() const { return type_ == Bool ? b_ : (type_ == Number ? num_ != 0.0 : false); }
    std::string asString() const { return type_ == String ? str_ : std::string(); }

    const Json &operator[](const std::string &key) const
    {
        static const Json null_value;
        if (type_ != Object)
            return null_value;
        auto it = obj_.find(key);
        return it == obj_.end() ? null_value : it->second;
    }
    const Json &operator[](size_t i) const
    {
        static const Json null_value;
        if (type_ != Array || i >= arr_.size())
            return null_value;
        return arr_[i];
    }
    const Json &operator[](int i) const { return (*this)[static_cast<size_t>(i)]; }

    // throws std::runtime_error with a line number on malformed input
    static Json parse(const std::string &text)
    {
        Parser p{text, 0};
        Json v = p.value();
        p.skip();
        if (p.pos != text.size())
            p.fail("trailing characters");
        return v;
    }

private:
    Type type_;
    double num_;
    bool b_;
    std::string str_;
    std::vector<Json> arr_;
    std::map<std::string, Json> obj_;

    struct Parser
    {
        const std::string &s;
        size_t pos;

        [[noreturn]] void fail(const char *what) const
        {
            size_t line = 1;
            for (size_t i = 0; i < pos && i < s.size(); i++)
                if (s[i] == '\n')
                    line++;
            throw std::runtime_error(std::string("json: ") + what + " at line " + std::to_string(line));
        }
        void skip()
        {
            for (;;)
            {
                while (pos < s.size() && (s[pos] == ' ' || s[pos] == '\t' || s[pos] == '\n' || s[pos] == '\r'))
                    pos++;
                if (pos + 1 < s.size() && s[pos] == '/' && s[pos + 1] == '/')
                {
                    while (pos < s.size() && s[pos] != '\n')
                        pos++;
                }
                else if (pos + 1 < s.size() && s[pos] == '/' && s[pos + 1] == '*')
                {
                    pos += 2;
                    while (pos + 1 < s.size() && !(s[pos] == '*' && s[pos + 1] == '/'))
                        pos++;
                    if (pos + 1 >= s.size())
                        fail("unterminated comment");
                    pos += 2;
                }
                else
                    return;
            }
        }
        Json value()
        {
            skip();
            if (pos >= s.size())
                fail("unexpected end of input");
            char c = s[pos];
            if (c == '{')
                return object();
            if (c == '[')
                return array();
            if (c == '"')
            {
                Json j;
                j.type_ = String;
                j.str_ = string();
                return j;
            }
            if (s.compare(pos, 4, "true") == 0)
            {
                pos += 4;
                Json j;
                j.type_ = Bool;
                j.b_ = true;
                return j;
            }
            if (s.compare(pos, 5, "false") == 0)
            {
                pos += 5;
                Json j;
                j.type_ = Bool;
                j.b_ = false;
                return j;
            }
            if (s.compare(pos, 4, "null") == 0)
            {
                pos += 4;
                return Json();
            }
            if (c == '-' || (c >= '0' && c <= '9'))
            {
                const char *b = s.c_str() + pos;
                char *e = nullptr;
                double v = std::strtod(b, &e);
                if (e == b)
                    fail("bad number");
                pos += static_cast<size_t>(e - b);
                Json j;
                j.type_ = Number;
                j.num_ = v;
                return j;
            }
            fail("unexpected character");
        }
        std::string string()
        {
            std::string out;
            pos++; // opening quote
            while (pos < s.size() && s[pos] != '"')
            {
                char c = s[pos++];
                if (c == '\\')
                {
                    if (pos >= s.size())
                        fail("bad escape");
                    char e = s[pos++];
                    switch (e)
                    {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u':
                    {
                        if (pos + 4 > s.size())
                            fail("bad \\u escape");
                        unsigned cp = static_cast<unsigned>(std::strtoul(s.substr(pos, 4).c_str(), nullptr, 16));
                        pos += 4;
                        if (cp < 0x80)
                            out += static_cast<char>(cp);
                        else if (cp < 0x800)
                        {
                            out += static_cast<char>(0xC0 | (cp >> 6));
                            out += static_cast<char>(0x80 | (cp & 0x3F));
                        }
                        else
                        {
                            out += static_cast<char>(0xE0 | (cp >> 12));
                            out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F));
                            out += static_cast<char>(0x80 | (cp & 0x3F));
                        }
                        break;
                    }
                    default: out += e; break;
                    }
                }
                else
                    out += c;
            }
            if (pos >= s.size())
                fail("unterminated string");
            pos++; // closing quote
            return out;
        }
        Json array()
        {
            Json j;
            j.type_ = Array;
            pos++;
            skip();
            if (pos < s.size() && s[pos] == ']')
            {
                pos++;
                return j;
            }
            for (;;)
            {
                j.arr_.push_back(value());
                skip();
                if (pos >= s.size())
                    fail("unterminated array");
                if (s[pos] == ',')
                {
                    pos++;
                    continue;
                }
                if (s[pos] == ']')
                {
                    pos++;
                    return j;
                }
                fail("expected , or ]");
            }
        }
        Json object()
        {
            Json j;
            j.type_ = Object;
            pos++;
            skip();
            if (pos < s.size() && s[pos] == '}')
            {
                pos++;
                return j;
            }
            for (;;)
            {
                skip();
                if (pos >= s.size() || s[pos] != '"')
                    fail("expected key string");
                std::string key = string();
                skip();
                if (pos >= s.size() || s[pos] != ':')
                    fail("expected :");
                pos++;
                j.obj_[key] = value(); // duplicate key: last one wins
                skip();
                if (pos >= s.size())
                    fail("unterminated object");
                if (s[pos] == ',')
                {
                    pos++;
                    continue;
                }
                if (s[pos] == '}')
                {
                    pos++;
                    return j;
                }
                fail("expected , or }");
            }
        }
    };
};

} // namespace ptx
