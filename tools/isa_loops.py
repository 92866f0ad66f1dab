"""Instruction mix of the loops of one kernel in a hipcc -S listing: tools/isa_loops.py file.s mangled_name_fragment"""
import collections
import sys


def main(path, frag):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and frag in l.split(":")[0] and l.rstrip().endswith(")") is False and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end + 1]
    print("kernel lines", len(body), "scratch ops", sum("scratch_" in l for l in body))
    for h, l in enumerate(body):
        if "Loop Header" not in l:
            continue
        lab = l.split(":")[0]
        ends = [i for i, x in enumerate(body) if "s_cbranch" in x and x.split()[-1] == lab and i > h]
        if not ends:
            continue
        loop = body[h:ends[-1] + 1]
        ops = [x.split()[0] for x in loop if x.strip() and not x.strip().startswith((".", ";")) and not x.rstrip().endswith(":")]
        c = collections.Counter(ops)
        grp = lambda p: sum(v for k, v in c.items() if k.startswith(p))
        print(lab, "instr", len(ops), "valu", grp("v_"), "salu", grp("s_") - c["s_waitcnt"], "waitcnt", c["s_waitcnt"], "ds", grp("ds_"),
              "global", grp("global_"), "scratch", grp("scratch_"), "nop", c["s_nop"])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
