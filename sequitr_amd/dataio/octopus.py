"""OctopusData: reader for Octopus camera streams (sequitr/dataio/octopus.py), Python 3.

A stream ``<stem><n>.dth`` / ``<stem><n>.dat`` (n = 0, 1, ...): the ``.dth`` is text, one line per frame of
``Key: value`` pairs (``H``, ``W``, optional ``Bit_Depth``, ...; octopus.py:215-229); the ``.dat`` is the raw
frames, ``uint<Bit_Depth>`` of shape (frames, H, W), opened as a numpy memmap (octopus.py:231-237).
``stream[n]`` returns frame n as a float array (octopus.py:239-245, 181-185).  ``block(first, count)`` is the
addition the GPU front end uses: the raw integer frames, no float conversion on the host.
"""
import os
import re
import time

import numpy as np


class OctopusData(object):
    def __init__(self, filename, contiguous=True, header=False, verbose=False, timeout=60):
        self.data = None
        self.fileopen = -1
        self.framesize = -1
        self.filenum_to_framerange = {}
        self.currentfile = -1
        self.use_contig = contiguous
        self._header_only = header
        self._verbose = verbose
        self.filename = filename
        self.filelist = []
        self.num_frames = 0
        self._header_keys = {}
        self._header = []
        self.timeout = timeout                                 # octopus.py:77: files younger than this are skipped
        self.refresh()
        if not self.filelist:
            raise IOError('No settled Octopus files for {0:s} (younger than {1}s?)'.format(filename, self.timeout))
        self._open_header(self.filename + str(self.filelist[0]))
        self.framesize = (int(self.header(0)['H']), int(self.header(0)['W']))
        self._bit_depth = int(self.header(0)['Bit_Depth']) if 'Bit_Depth' in self.header(0) else 16

    @property
    def bit_depth(self):
        return self._bit_depth

    @property
    def header_keys(self):
        return self._header_keys

    def header(self, frame_num):
        return dict((self._header_keys[i], self._header[frame_num][i]) for i in range(len(self._header_keys)))

    def _find_file_range(self):                                # octopus.py:113-151
        datadir, stem = os.path.split(self.filename)
        self.filestem = stem
        try:
            files = os.listdir(datadir or '.')
        except (IOError, OSError):
            raise IOError('No files exist in directory: {0:s}'.format(datadir))
        filenums = []
        for f in files:
            m = re.match(re.escape(stem) + r'([0-9]*)\.dth$', f)
            if m and m.group(1) != '':
                filenums.append(int(m.group(1)))
        if not filenums:
            raise IOError('No Octopus stream with pattern {0:s} found.'.format(stem))
        s = sorted(filenums)
        if not self.use_contig:
            return s
        out = [s[0]]
        for i in range(1, len(s)):
            if s[i] != s[i - 1] + 1:
                break
            out.append(s[i])
        return out

    def refresh(self):                                         # octopus.py:265-305
        to_update = []
        for nf in self._find_file_range():
            last_modified = os.stat(self.filename + str(nf) + '.dth').st_mtime
            if nf not in self.filelist and (time.time() - last_modified) > self.timeout:
                to_update.append(nf)
        if not to_update:
            return False
        for nf in to_update:
            self.filelist.append(nf)
            self._open_header(self.filename + str(nf))
            n = len(self._header)
            self.filenum_to_framerange[nf] = (self.num_frames, self.num_frames + n - 1)
            self.num_frames += n
        return True

    def _open_header(self, filename):                          # octopus.py:215-229
        try:
            with open(filename + '.dth') as fh:
                lines = [l for l in fh.readlines() if l.strip()]
        except IOError:
            raise IOError(filename + ' is not a valid file')
        self._header = [re.findall(r'\S+:\s*(\S+)', line) for line in lines]
        self._header_keys = re.findall(r'(\w*)\s*:\s*', lines[0])

    def _open_file(self, filename, num_frames):                # octopus.py:231-237
        try:
            self.data = np.memmap(filename + '.dat', dtype='uint' + str(self.bit_depth), mode='r',
                                  shape=(num_frames, self.framesize[0], self.framesize[1]))
        except (IOError, OSError, ValueError):
            self.data = []
            raise IOError(filename + ' is not a valid file. Make sure the path to the images still exists!')
        self.fileopen = True

    def _select(self, abs_frame_num):
        for i in self.filelist:
            lo, hi = self.filenum_to_framerange[i]
            if lo <= abs_frame_num <= hi:
                if self.currentfile != i or self.data is None:
                    self.currentfile = i
                    self._open_header(self.filename + str(i))
                    if not self._header_only:
                        self._open_file(self.filename + str(i), len(self._header))
                return i, abs_frame_num - lo
        raise IndexError('frame {0} is outside the stream (0..{1})'.format(abs_frame_num, self.num_frames - 1))

    def __getitem__(self, abs_frame_num):                      # octopus.py:153-159: the frame as floats
        _, rel = self._select(int(abs_frame_num))
        if self._header_only:
            return np.array(())
        return np.array(self.data[rel, :, :], dtype='float')

    def info(self, abs_frame_num):
        _, rel = self._select(int(abs_frame_num))
        d = self.header(rel)
        d['N'] = abs_frame_num
        return d

    def block(self, first, count):
        """Raw integer frames [first, first+count) as one (count, H, W) array (crossing file boundaries)."""
        out = np.empty((count,) + tuple(self.framesize), dtype='uint' + str(self.bit_depth))
        k = 0
        while k < count:
            i, rel = self._select(first + k)
            lo, hi = self.filenum_to_framerange[i]
            n = min(count - k, hi - lo + 1 - rel)
            out[k:k + n] = self.data[rel:rel + n]
            k += n
        return out

    def __len__(self):
        return self.num_frames

    def to_array(self):                                        # octopus.py:308-313 (uint8, as upstream)
        image_data = np.zeros((len(self), self.framesize[0], self.framesize[1]), dtype='uint8')
        for i in range(len(self)):
            image_data[i, ...] = self[i]
        return image_data
