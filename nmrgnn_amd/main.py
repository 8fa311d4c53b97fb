"""Command line of the hot path's caller: ``eval-struct`` (nmrgnn/main.py:192-278).

Same arguments, options, CSV columns and per-phase timing line as the reference command.  Frames
of a trajectory are independent graphs, so they are concatenated ``--frames-per-batch`` at a time into
one device batch (one launch sequence per batch instead of one per frame) and their neighbour lists
are built on the GPU (ng_knn_graph); the reference evaluates frame by frame with a CPU neighbour search.  The training / hyper-parameter-search commands of the reference are out of scope
(control plane)."""
from __future__ import annotations

import csv
import os
import sys
import time

import click
import numpy as np


@click.group()
def main():
    pass


def _open_structure(struct_files):
    from .structure import Structure, read_pdb
    first = read_pdb(struct_files[0])
    frames = list(first.frames)
    for extra in struct_files[1:]:                  # md.Universe(topology, *trajectory pieces)
        s = read_pdb(extra)
        if s.n_atoms != first.n_atoms:
            raise ValueError(f"{extra}: {s.n_atoms} atoms, but {struct_files[0]} has {first.n_atoms}")
        frames.extend(s.frames)
    return Structure(first.names, first.resnames, first.resids, first.elements, frames)


def eval_structure(struct_files, output_csv, model_file=None, neighbor_number=16, stride=1,
                   frames_per_batch=32, keep_going=False, device=None, echo=print):
    """Predict shifts for every ``stride``-th frame and write the reference's CSV.  Returns the timing
    buckets in seconds."""
    if len(struct_files) == 0:
        raise ValueError('Must pass at least on structure file')
    import torch
    from .graph import frames_to_batch
    from .library import check_peaks, load_model
    from .structure import atoms_onehot

    model = load_model(model_file, device=device)
    u = _open_structure(struct_files)
    frame_ids = list(range(0, len(u), stride))
    atoms = atoms_onehot(u.elements)
    n = atoms.shape[0]
    model.build(atoms.shape[1])
    model.freeze()          # inference only: the engine keeps its packed weight images across the per-frame calls
    timing = {'Structure': 0.0, 'Model Inference (MI355X)': 0.0, 'Parsing': 0.0}
    rows = []
    for b0 in range(0, len(frame_ids), max(1, frames_per_batch)):
        chunk = frame_ids[b0:b0 + max(1, frames_per_batch)]
        t = time.perf_counter()
        model.build(atoms.shape[1])
        dev = model.engine.device
        batch = frames_to_batch(atoms, np.stack([u.frames[fr] for fr in chunk]), neighbor_number, device=dev)
        torch.cuda.synchronize(dev)
        timing['Structure'] += time.perf_counter() - t
        t = time.perf_counter()
        peaks = model(batch).cpu().numpy()
        peaks = peaks.reshape(len(chunk), n)
        conf = []
        for k in range(len(chunk)):
            try:
                conf.append(check_peaks(atoms, peaks[k]))
            except Warning as w:
                if not keep_going:
                    raise
                echo(f'frame {chunk[k]}: {w}')
                conf.append(np.zeros(n, dtype=bool))
        timing['Model Inference (MI355X)'] += time.perf_counter() - t
        t = time.perf_counter()
        for k, fr in enumerate(chunk):
            pk = np.round(peaks[k].astype(np.float64), 2)
            for i in range(n):
                rows.append((i, u.resnames[i], int(u.resids[i]), u.names[i], pk[i], bool(conf[k][i]),
                             float(fr), fr))
        timing['Parsing'] += time.perf_counter() - t
        echo('|'.join(f'{k}:{v:5.2f}s' for k, v in timing.items()))
    os.makedirs(os.path.dirname(os.path.abspath(output_csv)), exist_ok=True)
    with open(output_csv, 'w', newline='') as f:
        wr = csv.writer(f)
        wr.writerow(['index', 'residues', 'resids', 'names', 'peaks', 'confident', 'time', 'frame'])
        wr.writerows(rows)
    echo(f'You can now find your result in {output_csv}')
    return timing


@main.command(name='eval-struct')
@click.argument('struct-files', nargs=-1, type=click.Path(exists=True))
@click.argument('output-csv')
@click.option('--model-file', type=click.Path(exists=True), default=None,
              help='Model file. If not provided, baseline will be used.')
@click.option('--neighbor-number', default=16, help='The model specific size of neighbor lists')
@click.option('--stride', default=1, help='Stride for reading trajectory, if multiple frames are present')
@click.option('--frames-per-batch', default=32, help='Frames evaluated per device batch')
@click.option('--keep-going', is_flag=True, help='Report implausible-shift warnings instead of aborting')
def eval_struct(struct_files, output_csv, model_file, neighbor_number, stride, frames_per_batch, keep_going):
    '''Predict NMR chemical shifts with specific file'''
    eval_structure(struct_files, output_csv, model_file, neighbor_number, stride, frames_per_batch,
                   keep_going, echo=click.echo)


if __name__ == '__main__':
    main()
