"""CSVExporter with the reference's file format (exporter.go:34-91): a creation-date comment, a header
`name,name+2s,name-2s,...`, then one row per estimate `x_i, +N*sqrt(P_ii), -N*sqrt(P_ii)` printed with %f,
and a closing-date comment.  Host-side I/O only (out of the hot path); it exists so that the reference's
example programs can be reproduced on top of the engine with byte-comparable outputs."""
import datetime
import math
import os


class CSVExporter:
    def __init__(self, headers, filepath, filename, covar_bound=2.0):
        self.covar_bound = float(covar_bound)
        self.delimiter = ","
        self.path = os.path.join(filepath, filename)
        self._fh = open(self.path, "w")
        bhdr = "%.0fs" % covar_bound
        hdr = []
        for h in headers:
            if h.startswith("_"):          # non-covariance column (exporter.go:66-70)
                hdr.append(h[1:])
                continue
            hdr += [h, h + "+" + bhdr, h + "-" + bhdr]
        self._fh.write("# Creation date (UTC): %s\n%s\n" % (datetime.datetime.utcnow(), self.delimiter.join(hdr)))

    def write(self, state, covar):
        """CSVExporter.Write(est) for one filter: state [n], covariance [n][n]."""
        vals = []
        for i in range(len(state)):
            c = self.covar_bound * math.sqrt(covar[i][i])
            vals += ["%f" % state[i], "%f" % c, "%f" % (-1 * c)]
        self._fh.write(self.delimiter.join(vals) + "\n")

    def write_raw_ln(self, s):
        self._fh.write(s + "\n")

    def close(self):
        self.write_raw_ln("# Closing date (UTC): %s\n" % datetime.datetime.utcnow())
        self._fh.close()
