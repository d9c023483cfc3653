#!/usr/bin/env python3
"""Generates tests/golden/zip_golden.json: which member of a PinMame-style zip the reference's loader takes for which ROM chip
(DCSDecoder::LoadROMFromZipFile, DCSDecoder/DCSDecoderZipLoader.cpp:60-207), for the 800 seeded archives of
romkit.zip_recognition_archive.

That file cannot be COMPILED in this image (it includes <Windows.h>; a stand-in header would be no build of the reference), so the
loader's execution cannot be recorded.  What this script does instead: it reads the recognition LITERALS out of the reference's
source text as data -- the three regular expressions, the '2' and '0' + n digit tests, the U7/U6 special case, IsJUMP -- checks
that each stands where the loader's control flow (read by the builder) expects it, evaluates them with Python's `re` (the three
patterns use nothing ECMAScript and Python read differently) and writes the expected chip of every member.  The literals are
recorded in the JSON, with the SHA-256 of the source lines they came from.  This pins the TEXT of the rules, not the reference's
execution of them.  Build container only (needs /root/reference); the output is committed."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import romkit                                                # noqa: E402

SRC = "/root/reference/DCSDecoder/DCSDecoderZipLoader.cpp"


def c_unescape(lit):
    """the characters a C string literal (without its quotes) stands for"""
    return re.sub(r"\\(.)", lambda m: {"n": "\n", "t": "\t", "0": "\0"}.get(m.group(1), m.group(1)), lit)


def read_literals():
    lines = open(SRC, encoding="latin-1").read().split("\n")
    region = lines[59:207]                                   # :60-207
    text = "\n".join(region)
    lit = {"source": "DCSDecoder/DCSDecoderZipLoader.cpp:60-207", "source_sha256": hashlib.sha256(text.encode("latin-1")).hexdigest()}
    # IsJUMP (:53)
    m = re.search(r"static inline bool IsJUMP\(const uint8_t \*p\) \{ (return [^}]*;) \}", "\n".join(lines[:60]))
    lit["jump_test"] = m.group(1)
    assert lit["jump_test"] == "return (p[0] & 0xFC) == 0x18 && (p[2] & 0x0F) == 0x0F;"
    # U2: IsJUMP(...) && strchr(filename, '<digit>'), or the explicitly named member (case-insensitive)
    m = re.search(r"IsJUMP\(rd\.data\.get\(\)\) && strchr\(rd\.filename\.c_str\(\), '(.)'\) != nullptr\)\s*\|\| \(explicitU2 != nullptr && _stricmp\(", text)
    lit["u2_digit"] = m.group(1)
    # the base name of the zip: path removed
    m = re.search(r'std::regex_replace\(zipFileName, std::regex\("((?:[^"\\]|\\.)*)"\), ""\)', text)
    lit["basename_regex_c_literal"] = m.group(1)
    lit["basename_regex"] = c_unescape(m.group(1))
    # U3..U9: for (int n = 3 ; n <= 9 ; ++n) ... desiredDigit = '0' + n ... strchr(filename, desiredDigit) ... regex_match(data, m, pat)
    m = re.search(r"for \(int n = (\d) ; n <= (\d) ; \+\+n\)", text)
    lit["chips"] = [int(m.group(1)), int(m.group(2))]
    assert re.search(r"char desiredDigit = '0' \+ n;", text) and re.search(r"rd\.chipNum < 0 && strchr\(rd\.filename\.c_str\(\), desiredDigit\) != nullptr", text)
    m = re.search(r'std::regex pat\("((?:[^"\\]|\\.)*)"\);\s*bool isMatch = std::regex_match\(reinterpret_cast<const char\*>\(rd\.data\.get\(\)\), m, pat\);', text)
    lit["signature_regex_c_literal"] = m.group(1)
    lit["signature_regex"] = c_unescape(m.group(1))
    assert re.search(r"char signatureDigit = isMatch \? m\[2\]\.str\(\)\.c_str\(\)\[0\] : 0;\s*bool load = \(signatureDigit == desiredDigit\);", text)
    # Cactus Canyon: zip base name matches <regex> (icase), a signature matched, chip '7' asked for, signature says '6'
    m = re.search(r'std::regex_match\(romZipFileBase, std::regex\("((?:[^"\\]|\\.)*)", std::regex_constants::icase\)\)\s*'
                  r"&& isMatch && desiredDigit == '(.)' && signatureDigit == '(.)'\)\s*load = true;", text)
    lit["cactus_canyon_regex_c_literal"] = m.group(1)
    lit["cactus_canyon_regex"] = c_unescape(m.group(1))
    lit["cactus_canyon_chip"], lit["cactus_canyon_signature_digit"] = m.group(2), m.group(3)
    return lit


def expected(members, zip_base, lit):
    """the loader's control flow (:127-204) over the recorded literals -> {chip: member index} or None (no U2)"""
    def is_jump(d):
        return len(d) >= 3 and (d[0] & 0xFC) == 0x18 and (d[2] & 0x0F) == 0x0F
    chip_of, taken = {}, set()
    for i, (name, data) in enumerate(members):
        if is_jump(data) and lit["u2_digit"] in name:
            chip_of[2] = i
            taken.add(i)
            break
    if 2 not in chip_of:
        return None
    base = re.sub(lit["basename_regex"], "", zip_base, count=1)
    sig = re.compile(lit["signature_regex"].encode("latin-1"))          # (std::regex: '.' stops at line terminators, as here)
    cactus = re.fullmatch(lit["cactus_canyon_regex"], base, re.I) is not None
    for n in range(lit["chips"][0], lit["chips"][1] + 1):
        want = str(n)
        for i, (name, data) in enumerate(members):
            if i in taken or want not in name:
                continue
            text = data.split(b"\0", 1)[0]                              # (regex_match on a const char *: up to the first NUL)
            m = sig.fullmatch(text)
            digit = m.group(2).decode() if m else ""
            load = digit == want
            if cactus and m is not None and want == lit["cactus_canyon_chip"] and digit == lit["cactus_canyon_signature_digit"]:
                load = True
            if load:
                chip_of[n] = i
                taken.add(i)
                break
    return chip_of


def main():
    lit = read_literals()
    archives = []
    for seed in range(800):
        arch = romkit.zip_recognition_archive(seed)
        if arch is None:
            continue
        members, zip_base, _ = arch
        chips = expected(members, zip_base, lit)
        archives.append(dict(seed=seed, zip_base=zip_base, members=[m[0] for m in members],
                             chips=None if chips is None else {str(c): i for c, i in sorted(chips.items())}))
    out = dict(note="expected chip -> member index per archive, from the recognition literals read out of the reference's source text "
                    "(not from its execution: the file includes <Windows.h> and cannot be built here)",
               literals=lit, archives=archives)
    with open(os.path.join(ROOT, "tests", "golden", "zip_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    n_u2 = sum(1 for a in archives if a["chips"] is not None)
    print("%d archives, %d with a U2, %d chips assigned, %d by the Cactus Canyon rule's zip names" %
          (len(archives), n_u2, sum(len(a["chips"]) for a in archives if a["chips"]), sum(1 for a in archives if a["zip_base"].lower().startswith("cc_"))))


if __name__ == "__main__":
    main()
