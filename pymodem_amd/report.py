"""Text reports of a decode run: the reference's `decoded_headers` report and its defect list
(PacketMetaArray.PrintRawBad / Report and print_ax25_header_to_string, packet_meta.py:43-169,283-370), character for
character (tests/test_report.py compares with text captured from the reference).  Control plane only: nothing here
touches samples.  Style "raw" raises NameError in the reference (packet_meta.py:328-329); here it prints what that
branch evidently meant to print."""

_U_FRAMES = {0x6F: "SABME", 0x2F: "SABM", 0x43: "DISC", 0x0F: "DM", 0x63: "UA", 0x87: "FRMR", 0x03: "UI", 0xAF: "XID", 0xE3: "TEST"}
_PIDS = {0x01: "ISO 8208", 0x06: "Compressed TCP/IP", 0x07: "Uncompressed TCP/IP", 0x08: "Segmentation Fragment", 0xC3: "TEXNET",
         0xC4: "Link Quality Protocol", 0xCA: "Appletalk", 0xCC: "ARPA Internet Protocol", 0xCD: "ARPA Address Resolution",
         0xCF: "TheNET (NET/ROM)", 0xF0: "No Layer 3", 0xFF: "Escape"}


def ax25_header_text(frame, delimiter):
    """-> (index of the first payload byte, header line).  Frames of 15 bytes or fewer give (0, '')."""
    count = len(frame)
    if count <= 15:
        return 0, ""
    out = []
    index = 0
    field = 0
    pos = 0
    extended = False
    while not extended and index < count:            # address fields until the extension bit
        ch = int(frame[index])
        if ch & 1:
            extended = True
        ch >>= 1
        pos += 1
        if pos == 1:
            out.append("To:" if field == 0 else delimiter + ("From:" if field == 1 else "Via:"))
        if pos < 7:
            if ch != 0 and ch != 0x20:
                out.append(chr(ch))
        elif pos == 7:
            out.append("-" + str(ch & 0b1111))
            if ch & 0b10000000:                      # never true after the shift of an 8-bit value; kept for fidelity
                out.append("* ")
            pos = 0
            field += 1
        index += 1
    if index < count:
        ctl = int(frame[index])
        out.append(delimiter + "Control: " + f"{hex(ctl)} ")
        frame_type = (ctl & 3) if (ctl & 1) else 0
        u_type = (ctl & 0xEF) if frame_type == 3 else 0
        out.append(_U_FRAMES.get(u_type, ""))
        if frame_type == 0 or u_type == 3:
            index += 1
            pid = int(frame[index])                  # IndexError on a truncated frame, like the reference
            out.append(delimiter + "PID: " + f"{hex(pid)} " + _PIDS.get(pid, ""))
        index += 1
    out.append(" \n")
    return index, "".join(out)


def _payload_text(data, start):
    return "".join(chr(int(b)) if 0x1F < int(b) < 0x7F else f"<{hex(int(b))}>" for b in data[start:len(data) - 2])


def raw_bad_text(arr):
    """PacketMetaArray.PrintRawBad, packet_meta.py:283-306."""
    out = []
    n = 0
    for packets in arr.raw_packet_arrays:
        for p in packets:
            if (p.ValidCRC is False) or (p.ValidHeader is False):
                n += 1
                why = (" bad CRC" if p.ValidCRC is False else "") + (" bad header" if p.ValidHeader is False else "")
                out.append(f"Frame with defect: {why}\n")
                out.append(f"Packet number:  {n} Calc CRC:  {hex(p.CalculatedCRC)} Carried CRC:  {hex(p.CarriedCRC)} stream address:  {p.streamaddress}\n")
                out.append(f"source decoder:  {p.SourceDecoder}\n")
                out.append(f"Packet byte count:  {len(p.data)}\n")
                out.append(f"Bytes corrected:  {p.BytesCorrected}\n")
                start, header = ax25_header_text(p.data, ", ")
                out.append(header + _payload_text(p.data, start) + "\n\n")
    arr.bad_count = n
    return "".join(out)


def report_text(arr, style):
    """PacketMetaArray.Report, packet_meta.py:337-370."""
    out = []
    if style.style == "decoded_headers":
        n = 0
        for p in arr.unique_packet_array:
            if p.ValidCRC and p.ValidHeader:
                n += 1
                out.append(f"\n\nPacket number:  {n}  CRC:  {hex(p.CalculatedCRC)} stream address:  {p.streamaddress}\n")
                out.append(f"Source decoders:  {p.CorrelatedDecoders}\n")
                out.append(f"Packet byte count:  {len(p.data)}\n")
                out.append(f"Bytes corrected:  {p.BytesCorrected}\n")
                start, header = ax25_header_text(p.data, ", ")
                out.append(header + _payload_text(p.data, start))
        out.append(f"\n\nUnique, valid packets:  {arr.CountGood()}\n")
        out.append(f"Packets rejected from all decoders for CRC failure:  {arr.CountBad()}\n")
        out.append("Total packets by decoder:\n")
        out.extend(f"{d} {c}\n" for d, c in arr.DecoderHistogram.most_common())
        out.append("Unique packets by decoder:\n")
        out.extend(f"{d} {c}\n" for d, c in arr.DecoderUniqueHistogram.most_common())
    elif style.style == "raw":
        out.append(raw_bad_text(arr))
        n = 0
        for p in arr.unique_packet_array:
            if p.ValidCRC and p.ValidHeader:
                n += 1
                out.append(f"Packet number:  {n}  CRC:  {hex(p.CalculatedCRC)} stream address:  {p.streamaddress}\n")
                out.append(f"source decoders:  {p.CorrelatedDecoders}\n")
                out.append(_payload_text(list(p.data) + [0, 0], 0) + " \n")
        out.append(f"\nValid packets:  {arr.CountGood()}\n")
        out.append(f"CRC saves:  {arr.CountBad()}\n")
    return "".join(out)
