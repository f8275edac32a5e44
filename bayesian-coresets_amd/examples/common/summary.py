"""Text / CSV summaries of stored runs: the role of the reference's Bokeh front end
(examples/common/plotting.py), which is UI and out of scope (bokeh is not in this image).  The grouping
semantics are kept: `--summarize` drops argument columns from the match, `--plot_legend` splits the
result into one series per value, `--groupby` reduces each series to 25/50/75 % percentiles per group."""
import numpy as np


def summarize(arguments, table, out=None):
    x, y = arguments.plot_x, arguments.plot_y
    legend = getattr(arguments, "plot_legend", None)
    groupby = getattr(arguments, "groupby", None)
    series = [(None, table)] if legend is None else list(table.groupby(legend))
    lines = []
    for name, t in series:
        if groupby is not None:
            g = t.groupby(groupby)
            rows = [(np.percentile(v[x], 50), np.percentile(v[y], 25), np.percentile(v[y], 50),
                     np.percentile(v[y], 75)) for _, v in g]
            hdr = "%s,%s_p25,%s_p50,%s_p75" % (x, y, y, y)
        else:
            rows = list(zip(t[x].to_numpy(), t[y].to_numpy()))
            hdr = "%s,%s" % (x, y)
        lines.append("# %s%s" % ("" if legend is None else "%s=%s  " % (legend, name), hdr))
        lines += [",".join("%.10g" % v for v in r) for r in sorted(rows)]
    text = "\n".join(lines)
    if out:
        with open(out, "w") as f:
            f.write(text + "\n")
    else:
        print(text)
    return text
