"""r/ccgp.R cannot run in this container (no R).  Its wrappers around the scripts' own compare.GP / prediction / factors.frame
assume things about those scripts; this test READS the eight reference scripts (text only -- nothing is copied, nothing is
executed) where /root/reference exists and checks every assumption:
  * every `a$name` the compare.GP / prediction wrappers read is a formal argument of that script's own definition
    (the wrappers bind arguments with match.call against those formals);
  * each script's `prediction` body contains exactly ONE `apply(pars.frame, 1, predict.post` -- the call the wrapper answers
    from the batched table by swapping `apply` in the function's environment;
  * the leading columns factors.frame builds are (p, theta1, theta2[, lambda]), then beta: what .ccgp.layout / the shim's
    LAYOUT_* and `pars[o + 1] is beta` rely on (HX:631-643, ANI:572-592, D1:750-781);
  * logpost / factors / predict.post / Mixed.corr.* signatures match the definitions r/ccgp.R overrides them with.
Skipped where the reference is absent (the GPU box)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SCRIPTS = {
    "HX": "Heat Exchanger Emulator/Combined GP Heat Exchanger.R",
    "GV": "Ground Vibrations Emulator/Combined GP Ground Vibrations.R",
    "ISO": "2D Codes and Designs/2D Combined GP Isotropic Public.R",
    "ADV": "2D Codes and Designs/2D Combined GP Isotropic Advanced.R",
    "ANI": "2D Codes and Designs/2D Combined GP Anisotropic Public.R",
    "BSQ": "Batch Sequential ME Designs/Batch Sequential ME Design.R",
    "D1": "1D Codes and Designs/1D Combined GP Public.R",
    "D1F": "1D Codes and Designs/1D Combined GP Two Families Public.R",
}

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference scripts not present on this machine")


def script(tag):
    with open(os.path.join(REF, SCRIPTS[tag]), encoding="latin-1") as fh:
        return fh.read()


def definition(src, name):
    """(formals, body) of the LAST top-level `name <- function(...) {...}` in an R source text."""
    hits = list(re.finditer(r"^%s\s*<-\s*function\s*\(" % re.escape(name), src, flags=re.M))
    if not hits:
        return None
    pos = hits[-1].end()
    depth, i = 1, pos
    while depth:
        depth += {"(": 1, ")": -1}.get(src[i], 0)
        i += 1
    formals = [a.strip().split("=")[0].strip() for a in src[pos:i - 1].replace("\n", " ").split(",")]
    j = src.index("{", i)
    depth, k = 1, j + 1
    while depth:
        depth += {"{": 1, "}": -1}.get(src[k], 0)
        k += 1
    return formals, src[j:k]


def wrapper_reads(rsrc, name):
    """names read as a$<name> inside r/ccgp.R's wrapper `name <- function(...) {`"""
    _, body = definition(rsrc, name)
    return set(re.findall(r"\ba\$([A-Za-z.][A-Za-z0-9._]*)", body))


@pytest.fixture(scope="module")
def rsrc():
    with open(os.path.join(ROOT, "r", "ccgp.R")) as fh:
        src = fh.read()
    # the wrappers live inside `if (exists(...)) { ... }` blocks: dedent them so that `definition` sees top-level assignments
    return re.sub(r"^  (compare\.GP|prediction|factors\.frame) <- function", r"\1 <- function", src, flags=re.M)


@pytest.mark.parametrize("tag", sorted(SCRIPTS))
def test_wrapped_arguments_are_formals_of_the_script(rsrc, tag):
    src = script(tag)
    for fn, alternatives in (("compare.GP", {"D.test": "D.new", "D.new": "D.test"}), ("prediction", {})):
        d = definition(src, fn)
        if d is None:
            assert tag == "BSQ" and False, "%s has no %s" % (tag, fn)
        formals = set(d[0])
        for name in wrapper_reads(rsrc, fn):
            if name == "nu":                 # read through .ccgp.nu(a): NULL where the script has no nu
                continue
            ok = name in formals or alternatives.get(name) in formals
            assert ok, "%s: the %s wrapper reads a$%s, formals are %s" % (tag, fn, name, sorted(formals))
        if tag in ("D1", "D1F"):
            assert "nu" in formals
        else:
            assert "nu" not in formals


@pytest.mark.parametrize("tag", sorted(SCRIPTS))
def test_prediction_has_exactly_one_apply_over_the_frame(tag):
    formals, body = definition(script(tag), "prediction")
    calls = re.findall(r"apply\s*\(\s*pars\.frame\s*,\s*1\s*,\s*predict\.post", body)
    assert len(calls) == 1, "%s: %d apply(pars.frame, 1, predict.post ...) calls" % (tag, len(calls))
    assert len(re.findall(r"\bapply\s*\(", body)) == 1          # the wrapper swaps EVERY apply in that environment
    assert "pars.frame" in formals and "x.new" in formals and "D.train" in formals and "sigma2" in formals
    # what comes back is transposed into an S x 2 frame named mean / var: the 2 x S block the wrapper returns fits
    assert re.search(r"data\.frame\s*\(\s*t\s*\(\s*apply", body) and re.search(r'c\("mean",\s*"var"\)', body)


@pytest.mark.parametrize("tag", sorted(SCRIPTS))
def test_frame_columns_are_the_layout_the_shim_indexes(rsrc, tag):
    formals, body = definition(script(tag), "factors.frame")
    m = re.search(r'names\(samp\)\s*<-\s*c\(([^)]*)\)', body)
    cols = [c.strip().strip('"') for c in m.group(1).split(",")]
    want = ["p", "theta1", "theta2", "lambda"] if tag == "ANI" else ["p", "theta1", "theta2"]
    assert cols == want, (tag, cols)
    assert re.search(r"return\s*\(\s*cbind\s*\(\s*samp\s*,\s*beta\s*,\s*prediction\.factors\s*,\s*R\.Inv\s*\)\s*\)", body)
    # r/ccgp.R: layout 2 (four leading columns, anisotropic rates) for ANI only; ADV's frame is (p, theta1, theta2) like HX's
    lay = re.search(r"\.ccgp\.layout\s*<-\s*switch\(ccgp\.script,\s*([^)]*)\)", rsrc).group(1)
    table = dict((k.strip(), v.strip()) for k, v in (e.split("=") for e in lay.split(",") if "=" in e))
    default = [e.strip() for e in lay.split(",") if "=" not in e][0]
    got = table.get(tag, default)
    assert got == {"ANI": "2L", "D1": "3L", "D1F": "4L"}.get(tag, "0L"), (tag, got)
    # y.train and D.train are formals factors.frame hands on to factors(): the wrapper's attribute comes from there
    assert "y.train" in formals and "D.train" in formals
    assert re.search(r"factors\s*,\s*n\.train\s*=\s*n\.train\s*,\s*y\.train\s*=\s*y\.train", body.replace("\n", " "))


def test_where_each_predict_post_finds_beta_and_its_second_scale():
    """The shim's `pars[o + 1] is beta` with o = 4 for the anisotropic layout and 3 otherwise, and ADV's predict.post as written:
    third leading column used as lambda in theta1 * (1 + lambda) (ADV:672) although the frame stores theta2 there (ADV:639-641)
    -- the inconsistency LAYOUT_ADV_WRITTEN reproduces on request and the batched path resolves towards the training kernel."""
    shim = open(os.path.join(ROOT, "r", "ccgp_shim.c")).read()
    assert re.search(r"layout == LAYOUT_ANI \? 4 : 3;\s*/\* pars\[o \+ 1\] is beta", shim)
    for tag in sorted(SCRIPTS):
        _, body = definition(script(tag), "predict.post")
        col = 5 if tag == "ANI" else 4
        assert re.search(r"beta\s*<-\s*as\.numeric\(pars\[%d\]\)" % col, body), (tag, col)
    _, adv = definition(script("ADV"), "predict.post")
    assert re.search(r"lambda\s*<-\s*as\.numeric\(pars\[3\]\)", adv) and "theta1*(1+lambda)" in adv.replace(" ", "")
    assert re.search(r"layout == LAYOUT_ADV_WRITTEN \? t1 \* \(1\.0 \+ t2\) : t2", shim)


@pytest.mark.parametrize("tag", sorted(SCRIPTS))
def test_overridden_signatures_match(tag):
    """The argument lists r/ccgp.R defines for this script's variant against the script's own."""
    src = script(tag)
    want = {
        "logpost": {"HX": 6, "ADV": 6, "D1": 5, "D1F": 5}.get(tag, 4),
        "factors": 3,
        "predict.post": 5 if tag in ("D1", "D1F") else 4,
    }
    for fn, nargs in want.items():
        d = definition(src, fn)
        assert d is not None, (tag, fn)
        assert len(d[0]) == nargs, (tag, fn, d[0])
    f, _ = definition(src, "factors")
    assert f == ["MCMC.data", "n.train", "y.train"]
    f, _ = definition(src, "predict.post")
    assert f[:4] == ["x.new", "D.train", "pars", "sigma2"]


@pytest.mark.parametrize("tag", ["HX", "GV", "ISO", "ADV", "ANI", "BSQ"])
def test_metro_loop_is_what_the_block_wise_metro_assumes(rsrc, tag):
    """r/ccgp.R's Metro pre-draws m iterations' random numbers.  That is the script's chain only if the script's loop draws
    exactly  u <- runif(1)  and then  rmnorm(1, as.vector(theta.old), sqrt(2)*pars$v)  per iteration, in this order, accepts on
    R > log(u) with R = l.cand$val - l.old$val, and runs Geweke's test on the first column of the last samp.size accepted draws
    whenever (k-1) >= samp.size & (k-1) %% batch.size == 0 -- and if every call site passes Metro's arguments by position."""
    src = script(tag)
    formals, body = definition(src, "Metro")
    flat = re.sub(r"\s+", "", body)
    assert formals[:8] == ["start", "N", "samp.size", "batch.size", "alpha", "D.train", "sigma2", "y"]
    assert len(formals) == (10 if tag in ("HX", "ADV") else 9), formals
    i_u, i_c = flat.index("u<-runif(1)"), flat.index("theta.candidate<-rmnorm(1,as.vector(theta.old),sqrt(2)*pars$v)")
    assert 0 < i_u < i_c and flat.count("runif(") == 1 and flat.count("rmnorm(") == 1 and "rnorm(" not in flat.replace("rmnorm(", "")
    assert "R<-l.cand$val-l.old$val" in flat and "if(R>log(u))" in flat
    assert "while(k<=N&pv<alpha)" in flat
    assert "if((k-1)>=samp.size&(k-1)%%batch.size==0)" in flat
    assert "geweke.diag(mcmc(samp[(k-samp.size):(k-1)]))$z" in flat
    assert ("est<-laplace(logpost.val,start)" in flat or "est<-laplace(logpost.val,start,...)" in flat) and "pars<-list(mu=est$mode,v=est$var)" in flat
    assert "theta.old<-pars$mu" in flat and "k=1" in flat
    tail = "R.Inv=R.Inv[(k-samp.size):(k-1)]" + (",logpost=log.posterior[(k-samp.size):(k-1)]" if tag == "BSQ" else "") + "))"
    assert flat.rstrip("}").endswith(tail)        # the list r/ccgp.R's Metro returns (it always adds BSQ's `logpost`)
    assert "logpost = log.posterior[(k - samp.size):(k - 1)]" in open(os.path.join(ROOT, "r", "ccgp.R")).read()
    # every call of Metro in the script passes its arguments by position
    for m in re.finditer(r"\bMetro\s*\(([^)]*)\)", src):
        if "function" in src[max(0, m.start() - 12):m.start()]:
            continue
        assert "=" not in m.group(1), m.group(0)
    # r/ccgp.R's replacement takes the same number of arguments
    r_formals = re.search(r"Metro <- function\(([^)]*)\)\s*\n\s*\.ccgp\.Metro\(start, N, samp\.size, batch\.size, alpha, D\.train, sigma2, y,\s*\n\s*c\(\.ccgp\.prior",
                          open(os.path.join(ROOT, "r", "ccgp.R")).read())
    assert r_formals and len(r_formals.group(1).split(",")) == 10
